"""Model files (`-d/--modelfile`) without scikit-learn.

The reference unpickles a scikit-learn estimator, or a dict of them keyed by sub-model, and calls
`predict_proba` on it (extract_contexts.py:123-130, :199).  Here the pickle is read with a restricted
unpickler -- every `sklearn.*` class becomes an inert attribute bag, only numpy array reconstruction is
allowed through -- and the arrays the forward pass needs are handed to the HIP classifier kernel.
"""
import io
import pickle

import numpy as np


class _Bag(object):
    """Stand-in for any scikit-learn class found in a model pickle."""

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):
            if isinstance(state[0], dict):
                self.__dict__.update(state[0])
            self.__dict__.update(state[1])
        else:
            self.__dict__['_state'] = state

    def __reduce_ex__(self, protocol):          # never re-pickled
        raise TypeError('model stand-ins are read-only')


_ALLOWED = {
    ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
    ('numpy', 'ndarray'), ('numpy', 'dtype'),
    ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
    ('numpy.random._pickle', '__randomstate_ctor'), ('numpy.random._pickle', '__bit_generator_ctor'),
    ('copy_reg', '_reconstructor'), ('copyreg', '_reconstructor'),
    ('__builtin__', 'object'), ('builtins', 'object'),
    ('collections', 'OrderedDict'), ('collections', 'defaultdict'),
    ('_codecs', 'encode'),
}


class _ModelUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        root = module.split('.')[0]
        if root == 'sklearn':
            return type(str(name), (_Bag,), {'_sk_module': module})
        if (module, name) in _ALLOWED:
            return pickle.Unpickler.find_class(self, module, name)
        if root == 'numpy' and name in ('_reconstruct', 'ndarray', 'dtype', 'scalar'):
            return pickle.Unpickler.find_class(self, module, name)
        if root == 'numpy' and module.startswith('numpy.random'):
            return type(str(name), (_Bag,), {'_sk_module': module})
        raise pickle.UnpicklingError('model file refers to %s.%s, which is not allowed' % (module, name))


class MLPWeights(object):
    """7 -> H -> 1 perceptron as scikit-learn's MLPClassifier stores it."""
    kind = 'mlp'

    def __init__(self, W1, b1, W2, b2, activation='tanh', classes=None):
        self.W1 = np.ascontiguousarray(W1, dtype=np.float64)
        self.b1 = np.ascontiguousarray(b1, dtype=np.float64)
        self.W2 = np.ascontiguousarray(W2, dtype=np.float64).reshape(-1)
        self.b2 = np.ascontiguousarray(b2, dtype=np.float64).reshape(-1)
        self.activation = activation
        self.classes = classes
        self.n_in, self.n_hidden = self.W1.shape


class ForestWeights(object):
    """scikit-learn RandomForestClassifier (binary): flattened trees.  Node arrays are concatenated over trees; leaves have
    left == -1; `value` holds the two class values of every node as the tree stores them (predict_proba normalises)."""
    kind = 'forest'

    def __init__(self, trees, n_features, classes=None):
        off, left, right, feat, thr, val = [0], [], [], [], [], []
        for nodes, values in trees:
            base = off[-1]
            n = len(nodes)
            l = nodes['left_child'].astype(np.int64)
            r = nodes['right_child'].astype(np.int64)
            left.append(np.where(l >= 0, l + base, -1))
            right.append(np.where(r >= 0, r + base, -1))
            feat.append(nodes['feature'].astype(np.int64))
            thr.append(nodes['threshold'].astype(np.float64))
            v = np.asarray(values, dtype=np.float64).reshape(n, -1)
            if v.shape[1] != 2:
                raise NotImplementedError('forests with %d classes' % v.shape[1])
            val.append(v)
            off.append(base + n)
        self.tree_off = np.asarray(off, dtype=np.int32)
        self.left = np.concatenate(left).astype(np.int32)
        self.right = np.concatenate(right).astype(np.int32)
        self.feature = np.concatenate(feat).astype(np.int32)
        self.threshold = np.ascontiguousarray(np.concatenate(thr), dtype=np.float64)
        self.value = np.ascontiguousarray(np.concatenate(val), dtype=np.float64)
        self.n_in = int(n_features)
        self.n_trees = len(trees)
        self.classes = classes


class LogisticWeights(object):
    """scikit-learn LogisticRegression, two classes (train_model.py:55-57, `-c LR`): p = expit(x . coef_ + intercept_)."""
    kind = 'logistic'

    def __init__(self, coef, intercept, classes=None):
        self.coef = np.ascontiguousarray(coef, dtype=np.float64).reshape(-1)
        self.intercept = float(np.asarray(intercept, dtype=np.float64).reshape(-1)[0])
        self.n_in = len(self.coef)
        self.classes = classes

    def params(self):
        return np.concatenate([self.coef, [self.intercept]])


class GaussianNBWeights(object):
    """scikit-learn GaussianNB, two classes (train_model.py:59-60, `-c NBC`): per class the feature means and variances (the
    variances as the estimator stores them, smoothing included) and the class prior."""
    kind = 'gnb'

    def __init__(self, theta, var, prior, classes=None):
        self.theta = np.ascontiguousarray(theta, dtype=np.float64)
        self.var = np.ascontiguousarray(var, dtype=np.float64)
        self.prior = np.ascontiguousarray(prior, dtype=np.float64).reshape(-1)
        if self.theta.shape != self.var.shape or self.theta.shape[0] != 2 or len(self.prior) != 2:
            raise NotImplementedError('naive Bayes models with %d classes' % self.theta.shape[0])
        self.n_in = self.theta.shape[1]
        self.classes = classes

    def params(self):
        return np.concatenate([self.theta[0], self.var[0], self.theta[1], self.var[1], np.log(self.prior)])


def _as_text(x):
    return x.decode('latin1') if isinstance(x, bytes) else str(x)


def _estimator_weights(est, where):
    cls = type(est).__name__
    if cls == 'RandomForestClassifier':
        trees = [(t.tree_.nodes, t.tree_.values) for t in est.estimators_]
        n_feat = int(getattr(est, 'n_features_in_', getattr(est, 'n_features_', 0)) or 0)
        return ForestWeights(trees, n_feat, [_as_text(c) for c in getattr(est, 'classes_', [])])
    if cls == 'LogisticRegression':
        coef = np.asarray(est.coef_, dtype=np.float64)
        if coef.ndim != 2 or coef.shape[0] != 1:
            raise NotImplementedError('%s: logistic regression with %s coefficients (two classes are supported)' % (where, coef.shape))
        # predict_proba of a binary model fitted with multi_class='multinomial' is softmax([-d, d])[1] = expit(2 d), not
        # expit(d) (scikit-learn < 1.7 keeps the attribute; the reference's own fit is 'ovr', train_model.py:56): twice the
        # coefficients give the same numbers through the one formula the kernel has
        scale = 2.0 if _as_text(getattr(est, 'multi_class', 'auto')) == 'multinomial' else 1.0
        return LogisticWeights(scale * coef[0], scale * np.asarray(est.intercept_, dtype=np.float64),
                               [_as_text(c) for c in getattr(est, 'classes_', [])])
    if cls == 'GaussianNB':
        var = getattr(est, 'var_', None)
        if var is None:
            var = est.sigma_                                # (scikit-learn < 1.0)
        return GaussianNBWeights(est.theta_, var, est.class_prior_, [_as_text(c) for c in getattr(est, 'classes_', [])])
    if cls != 'MLPClassifier':
        raise NotImplementedError('%s: classifier %s is not supported by the HIP path (MLPClassifier, RandomForestClassifier, '
                                  'LogisticRegression and GaussianNB are)' % (where, cls))
    coefs, inter = est.coefs_, est.intercepts_
    act = _as_text(getattr(est, 'activation', 'tanh'))
    out_act = _as_text(getattr(est, 'out_activation_', 'logistic'))
    if len(coefs) != 2 or coefs[1].shape[1] != 1 or out_act != 'logistic' or act != 'tanh':
        raise NotImplementedError('%s: only one-hidden-layer tanh/logistic MLPs are supported '
                                  '(got %d layers, %s/%s)' % (where, len(coefs), act, out_act))
    classes = [_as_text(c) for c in getattr(est, 'classes_', [])]
    return MLPWeights(coefs[0], inter[0], coefs[1], inter[1], act, classes)


class ModelSet(object):
    """What `model` is in the reference after :123-130: sub-model key -> estimator, plus `twobase`."""

    def __init__(self, models, twobase):
        self.models = models            # dict key -> MLPWeights, insertion ordered
        self.twobase = twobase

    def keys(self):
        return list(self.models.keys())


def load_model_file(path):
    """Pickle of an estimator or of a dict of estimators (the reference's format), or our neutral .npz export
    ('<key>.W1' ... arrays; a single key 'general' is read as a bare estimator, anything else as a dict)."""
    with open(path, 'rb') as fh:
        raw = fh.read()
    if raw[:2] == b'PK':                                    # numpy .npz
        files = np.load(path).files
        keys = sorted(set(n.split('.')[0] for n in files if not n.startswith('__')))
        return load_npz_weights(path, '__is_dict__' in files or keys != ['general'])
    obj = _ModelUnpickler(io.BytesIO(raw), encoding='latin1').load()
    if type(obj) != dict:                                   # extract_contexts.py:126-128
        return ModelSet({'general': _estimator_weights(obj, path)}, False)
    return ModelSet({_as_text(k): _estimator_weights(v, '%s[%s]' % (path, k)) for k, v in obj.items()}, True)


def shipped_model(stem='r95_twobase_model_NN_6_m6A'):
    """Path of a weight export shipped with the package (the reference keeps its .pkl models next to mCaller.py)."""
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), 'models', stem + '.npz')


def load_npz_weights(path, is_dict):
    """Neutral weight export (mcaller_amd/models/*.npz): arrays '<key>.W1' ... '<key>.b2'."""
    z = np.load(path)
    keys = sorted(set(n.split('.')[0] for n in z.files if not n.startswith('__')))
    models = {k: MLPWeights(z[k + '.W1'], z[k + '.b1'], z[k + '.W2'], z[k + '.b2'],
                            classes=[str(c) for c in z[k + '.classes']] if k + '.classes' in z.files else None) for k in keys}
    return ModelSet(models, is_dict)
