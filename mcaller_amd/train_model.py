"""`--train`: fit a classifier on the labelled feature matrix the GPU path built (train_model.py:33-113).

Only the feature-matrix build is on the accelerated path; the fit itself is the reference's recipe on
scikit-learn when that package is installed (class balancing :81-86, 5-fold GroupKFold by context :62-65,:92, fit, pickle
of {sub-model: estimator} :110-112).  `RF` drops `min_impurity_split`, which scikit-learn >= 1.0 no longer accepts."""
import pickle


def train_classifier(signals, groups, modelfile, classifier='NN', plot=False):
    try:
        from sklearn.ensemble import RandomForestClassifier
        from sklearn.linear_model import LogisticRegression
        from sklearn.model_selection import GroupKFold, cross_val_score
        from sklearn.naive_bayes import GaussianNB
        from sklearn.neural_network import MLPClassifier
        from sklearn import svm
    except ImportError:
        raise ImportError('--train needs scikit-learn for the fit (the feature matrix has been written to the .train file)')
    if plot:
        raise NotImplementedError('--plot_training is not supported')
    models = {}
    for twobase_model in signals:
        if classifier == 'RF':
            model = RandomForestClassifier(bootstrap=True, criterion='entropy', max_depth=10, max_features=4,
                                           min_samples_leaf=2, min_samples_split=3, n_estimators=50)
        elif classifier == 'NN':
            model = MLPClassifier(hidden_layer_sizes=(100), alpha=0.001, learning_rate='adaptive', early_stopping=False,
                                  activation='tanh')
        elif classifier == 'SVM':
            model = svm.SVC(kernel='rbf', probability=True)
        elif classifier == 'LR':
            model = LogisticRegression(solver='liblinear', penalty='l1')
        elif classifier == 'NBC':
            model = GaussianNB()
        else:
            raise ValueError('unknown classifier ' + str(classifier))
        num_examples = min([len(signals[twobase_model][label]) for label in signals[twobase_model]])
        labs, sigs, grps = [], [], []
        for label in signals[twobase_model]:
            labs = labs + [label] * num_examples
            sigs = sigs + signals[twobase_model][label][:num_examples]
            grps = grps + groups[twobase_model][label][:num_examples]
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
