"""`--train`: fit a classifier on the labelled feature matrix the GPU path built (train_model.py:33-113).

`NN` (the default, train_model.py:47) is fitted on the GPU: class balancing (:81-86), 5-fold GroupKFold by context
(:62-65, :92) and the final fit (:100) are six independent runs of the same optimiser, launched together as one
`mc_mlp_fit` call (four workgroups per fit, mcaller_amd/csrc/mc_train.hip).  The optimiser is scikit-learn's
MLPClassifier recipe (Adam, tanh, alpha=0.001, batches of min(200, n), tol/n_iter_no_change stopping); like the
reference's `random_state=None` fit, two runs differ unless MCALLER_SEED is set.

The model file is what the reference writes -- a pickle of {sub-model: MLPClassifier} (:110-112) -- when scikit-learn is
importable (the estimators are filled with the fitted arrays, so the reference can load them); otherwise a neutral
`.npz` of the same arrays that mcaller_amd.model_io reads.  The other classifiers (RF, SVM, LR, NBC) are fitted by
scikit-learn itself when it is installed (`RF` without `min_impurity_split`, which scikit-learn >= 1.0 rejects).
"""
import os
import pickle

import numpy as np


def pos2label(positions):
    """train_model.py:18-20."""
    return {(pos.split()[0], int(pos.split()[1]), pos.split()[2]): pos.split()[3]
            for pos in open(positions, 'r').read().split('\n') if len(pos.split()) > 1}


def group_kfold(groups, n_splits=5):
    """Fold of every sample under scikit-learn's GroupKFold (train_model.py:62-63): groups, largest first, are dealt to
    the currently lightest fold."""
    uniq, inv = np.unique(np.asarray(groups), return_inverse=True)
    if len(uniq) < n_splits:
        raise ValueError('Cannot have number of splits n_splits=%d greater than the number of groups: %d.'
                         % (n_splits, len(uniq)))
    per_group = np.bincount(inv)
    order = np.argsort(per_group)[::-1]
    per_fold = np.zeros(n_splits)
    group_to_fold = np.zeros(len(uniq), dtype=np.int64)
    for gi in order:
        lightest = int(np.argmin(per_fold))
        per_fold[lightest] += per_group[gi]
        group_to_fold[gi] = lightest
    return group_to_fold[inv]


def balanced_rows(signals, groups):
    """train_model.py:81-86: the first min-class-size rows of every label, label by label."""
    num_examples = min([len(signals[label]) for label in signals])
    labs, sigs, grps = [], [], []
    for label in signals:
        labs = labs + [label] * num_examples
        sigs = sigs + signals[label][:num_examples]
        grps = grps + groups[label][:num_examples]
    return labs, sigs, grps


def _seed():
    env = os.environ.get('MCALLER_SEED', '')
    return int(env) if env != '' else int.from_bytes(os.urandom(7), 'little')


def fit_nn_on_gpu(labs, sigs, grps, use_groups, device=None, hidden=100):
    """-> (classes, cross-validation scores, final weights dict)."""
    from .device import get_device
    dev = device if device is not None else get_device()
    classes = sorted(set(labs))                                   # LabelBinarizer order == estimator.classes_
    if len(classes) != 2:
        raise ValueError('training needs exactly two labels in the positions file, got %s' % classes)
    X = np.asarray(sigs, dtype=np.float64)
    y = np.asarray([1 if lab == classes[1] else 0 for lab in labs], dtype=np.uint8)
    n = len(y)
    if use_groups:
        fold = group_kfold(grps, 5)
    else:                                                         # cv=5 -> StratifiedKFold without shuffling
        fold = np.zeros(n, dtype=np.int64)
        for cls in (0, 1):
            rows = np.nonzero(y == cls)[0]
            fold[rows] = (np.arange(len(rows)) * 5) // max(len(rows), 1)
    rows = np.arange(n)
    jobs = [(rows[fold != f], rows[fold == f]) for f in range(5)] + [(rows, np.zeros(0, dtype=np.int64))]
    seed = _seed()
    fits = dev.mlp_fit(X, y, jobs, hidden=hidden, seeds=[(seed + 0x9E3779B97F4A7C15 * j) % (1 << 64) for j in range(6)])
    scores = np.array([f['val_correct'] / float(f['n_val']) for f in fits[:5]])
    return classes, scores, fits[5]


def as_sklearn_estimator(fit, classes, n_samples):
    """A scikit-learn MLPClassifier holding the fitted arrays (what the reference pickles, train_model.py:110-112)."""
    from sklearn.neural_network import MLPClassifier
    from sklearn.preprocessing import LabelBinarizer
    m = MLPClassifier(hidden_layer_sizes=(len(fit['b1'])), alpha=0.001, learning_rate='adaptive', early_stopping=False,
                      activation='tanh')
    m.coefs_ = [np.array(fit['W1'], dtype=np.float64), np.array(fit['W2'], dtype=np.float64).reshape(-1, 1)]
    m.intercepts_ = [np.array(fit['b1'], dtype=np.float64), np.array([fit['b2']], dtype=np.float64)]
    m.n_features_in_ = m.coefs_[0].shape[0]
    m.n_layers_, m.n_outputs_, m.out_activation_ = 3, 1, 'logistic'
    m.classes_ = np.array(classes)
    m._label_binarizer = LabelBinarizer().fit(classes)
    m.loss_curve_ = [float(x) for x in fit['loss_curve']]
    m.loss_ = m.loss_curve_[-1] if m.loss_curve_ else float('nan')
    m.best_loss_ = min(m.loss_curve_) if m.loss_curve_ else float('nan')
    m.n_iter_ = int(fit['n_iter'])
    m.t_ = int(fit['n_iter']) * int(n_samples)
    m.validation_scores_ = None
    m.best_validation_score_ = None
    return m


def write_models(models, classes_of, n_of, modelfile):
    try:
        import sklearn  # noqa: F401
        have_sklearn = True
    except ImportError:
        have_sklearn = False
    if have_sklearn:
        out = {key: as_sklearn_estimator(fit, classes_of[key], n_of[key]) for key, fit in models.items()}
        with open(modelfile, 'wb') as modfi:
            pickle.dump(out, modfi)
        return out
    arrays = {'__is_dict__': np.array([1])}
    for key, fit in models.items():
        arrays[key + '.W1'], arrays[key + '.b1'] = fit['W1'], fit['b1']
        arrays[key + '.W2'], arrays[key + '.b2'] = fit['W2'], np.array([fit['b2']])
        arrays[key + '.classes'] = np.array(classes_of[key])
    with open(modelfile, 'wb') as modfi:
        np.savez(modfi, **arrays)
    return models


def train_classifier(signals, groups, modelfile, classifier='NN', plot=False, device=None):
    if plot:
        raise NotImplementedError('--plot_training is not supported (it raises NameError in the reference: the import '
                                  'of plotlib is commented out, train_model.py:3,:108)')
    if classifier != 'NN':
        return _train_with_sklearn(signals, groups, modelfile, classifier)
    models, classes_of, n_of = {}, {}, {}
    for twobase_model in signals:
        labs, sigs, grps = balanced_rows(signals[twobase_model], groups[twobase_model])
        print(labs[:10])
        print(sigs[:10])
        print(grps[:10])
        classes, scores, fit = fit_nn_on_gpu(labs, sigs, grps, bool(groups), device=device)
        print('%s %s model scores: %s' % (classifier, twobase_model, ','.join([str(s) for s in scores])))
        print('Cross validation accuracy: %0.2f (+/- %0.2f)' % (scores.mean(), scores.std() * 2))
        models[twobase_model], classes_of[twobase_model], n_of[twobase_model] = fit, classes, len(labs)
    return write_models(models, classes_of, n_of, modelfile)


def _train_with_sklearn(signals, groups, modelfile, classifier):
    try:
        from sklearn.ensemble import RandomForestClassifier
        from sklearn.linear_model import LogisticRegression
        from sklearn.model_selection import GroupKFold, cross_val_score
        from sklearn.naive_bayes import GaussianNB
        from sklearn import svm
    except ImportError:
        raise ImportError('--train -c %s needs scikit-learn for the fit (only NN is fitted on the GPU; the feature matrix '
                          'has been written to the .train file)' % classifier)
    models = {}
    for twobase_model in signals:
        if classifier == 'RF':
            model = RandomForestClassifier(bootstrap=True, criterion='entropy', max_depth=10, max_features=4,
                                           min_samples_leaf=2, min_samples_split=3, n_estimators=50)
        elif classifier == 'SVM':
            model = svm.SVC(kernel='rbf', probability=True)
        elif classifier == 'LR':
            model = LogisticRegression(solver='liblinear', penalty='l1')
        elif classifier == 'NBC':
            model = GaussianNB()
        else:
            raise ValueError('unknown classifier ' + str(classifier))
        labs, sigs, grps = balanced_rows(signals[twobase_model], groups[twobase_model])
        print(labs[:10])
        print(sigs[:10])
        print(grps[:10])
        scores = cross_val_score(model, sigs, labs, cv=GroupKFold(n_splits=5) if groups else 5, groups=grps)
        print('%s %s model scores: %s' % (classifier, twobase_model, ','.join([str(s) for s in scores])))
        print('Cross validation accuracy: %0.2f (+/- %0.2f)' % (scores.mean(), scores.std() * 2))
        model.fit(sigs, labs)
        models[twobase_model] = model
    with open(modelfile, 'wb') as modfi:
        pickle.dump(models, modfi)
    return models
