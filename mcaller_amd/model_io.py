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


def _as_text(x):
    return x.decode('latin1') if isinstance(x, bytes) else str(x)


def _estimator_weights(est, where):
    cls = type(est).__name__
    if cls != 'MLPClassifier':
        raise NotImplementedError('%s: classifier %s is not supported by the HIP path yet '
                                  '(MLPClassifier only)' % (where, cls))
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
        keys = sorted(set(n.split('.')[0] for n in np.load(path).files))
        return load_npz_weights(path, keys != ['general'])
    obj = _ModelUnpickler(io.BytesIO(raw), encoding='latin1').load()
    if type(obj) != dict:                                   # extract_contexts.py:126-128
        return ModelSet({'general': _estimator_weights(obj, path)}, False)
    return ModelSet({_as_text(k): _estimator_weights(v, '%s[%s]' % (path, k)) for k, v in obj.items()}, True)


def load_npz_weights(path, is_dict):
    """Neutral weight export (tests/golden/models/*.npz): arrays '<key>.W1' ... '<key>.b2'."""
    z = np.load(path)
    keys = sorted(set(n.split('.')[0] for n in z.files))
    models = {k: MLPWeights(z[k + '.W1'], z[k + '.b1'], z[k + '.W2'], z[k + '.b2']) for k in keys}
    return ModelSet(models, is_dict)
