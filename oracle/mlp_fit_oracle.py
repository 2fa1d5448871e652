"""CPU restatement of the classifier fit behind `--train` (TEST INFRASTRUCTURE: only tests/, smoke() and bench.py's CPU
leg may import this; the product never does).

The reference fits `MLPClassifier(hidden_layer_sizes=(100), alpha=0.001, learning_rate='adaptive', early_stopping=False,
activation='tanh')` (train_model.py:47) on class-balanced rows (train_model.py:81-86) and scores it with 5-fold
GroupKFold by context (train_model.py:62-65,:92).  The arithmetic lives in scikit-learn (third-party, not under
/root/reference, no version pinned by the reference; 1.7.2 in the build container).  Restated here from its published
algorithm (sklearn/neural_network/_multilayer_perceptron.py, _stochastic_optimizers.py, model_selection/_split.py):

* one hidden tanh layer, logistic output, binary log-loss on probabilities clipped to [eps, 1-eps], plus
  0.5*alpha*sum(W^2)/n_batch;  gradients (a^T.delta + alpha*W)/n_batch and mean(delta);
* Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t), m/v moments, update -lr_t*m/(sqrt(v)+1e-8), t counted in batches;
* batches of min(200, n) rows in the epoch's order, the last one smaller; loss of an epoch = sum(batch_loss*n_batch)/n;
* stop when the loss failed to improve on the best by more than tol=1e-4 for more than 10 consecutive epochs, or after
  max_iter=200 epochs;
* Glorot-uniform start with bound sqrt(6/(fan_in+fan_out)) for coefficients and intercepts of a layer.

Pin: `tests/golden/make_golden_train.py` runs scikit-learn itself from the same start weights with shuffle=False and
commits its loss curve and final weights (tests/golden/train/); `tests/test_train.py` checks this file against them.
What is NOT pinned (and cannot be: the reference uses random_state=None): the random start and the epoch shuffles.
Those are ours -- a counter-based generator and a Feistel permutation, defined below and used identically by the HIP
kernel (mcaller_amd/csrc/mc_train.hip) so that the two can be compared step by step.
"""
import numpy as np

M64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def uniform01(seed, index):
    return (splitmix64((seed + index * 0xD1342543DE82EF95) & M64) >> 11) * (1.0 / 9007199254740992.0)


def init_weights(d, h, seed):
    """-> W1[d,h], b1[h], W2[h], b2 (float64).  Parameter index: W1 row-major, then b1, W2, b2."""
    b_hidden = np.sqrt(6.0 / (d + h))
    b_out = np.sqrt(6.0 / (h + 1))
    u = np.array([uniform01(seed, i) for i in range(d * h + 2 * h + 1)])
    W1 = (-b_hidden + 2.0 * b_hidden * u[:d * h]).reshape(d, h)
    b1 = -b_hidden + 2.0 * b_hidden * u[d * h:d * h + h]
    W2 = -b_out + 2.0 * b_out * u[d * h + h:d * h + 2 * h]
    b2 = -b_out + 2.0 * b_out * u[d * h + 2 * h]
    return W1, b1, W2, float(b2)


def _mix32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def epoch_key(seed, epoch):
    return splitmix64((seed ^ 0xA5A5A5A55A5A5A5A) + epoch) & 0xFFFFFFFF


def feistel_perm(i, n, key):
    """Bijection of [0, n): 4-round Feistel network on the next even power of two, cycle-walked into range."""
    bits = max(2, (n - 1).bit_length())
    bits += bits & 1
    half = bits // 2
    mask = (1 << half) - 1
    x = i
    while True:
        L, R = x >> half, x & mask
        for r in range(4):
            F = _mix32(R * 0x9E3779B1 + key + r * 0x85EBCA6B) & mask
            L, R = R, L ^ F
        x = (L << half) | R
        if x < n:
            return x


def epoch_order(n, seed, epoch, shuffle=True):
    if not shuffle:
        return np.arange(n)
    key = epoch_key(seed, epoch)
    return np.array([feistel_perm(i, n, key) for i in range(n)], dtype=np.int64)


def forward(W1, b1, W2, b2, X):
    a = np.tanh(X @ W1 + b1)
    z = a @ W2 + b2
    return a, 1.0 / (1.0 + np.exp(-z))


def fit(X, y, hidden=100, alpha=0.001, lr=0.001, beta1=0.9, beta2=0.999, eps=1e-8, batch_size=200, max_iter=200,
        tol=1e-4, n_iter_no_change=10, seed=1, shuffle=True, init=None):
    """-> dict(W1, b1, W2, b2, loss_curve, n_iter).  X float64 [n, d]; y in {0, 1}."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n, d = X.shape
    W1, b1, W2, b2 = init if init is not None else init_weights(d, hidden, seed)
    W1, b1, W2 = W1.copy(), b1.copy(), W2.copy()
    params = [W1, b1, W2, np.array([b2])]
    ms = [np.zeros_like(p) for p in params]
    vs = [np.zeros_like(p) for p in params]
    B = min(batch_size, n)
    t = 0
    best, no_improve = np.inf, 0
    curve = []
    feps = np.finfo(np.float64).eps
    for epoch in range(max_iter):
        order = epoch_order(n, seed, epoch, shuffle)
        acc = 0.0
        for b0 in range(0, n, B):
            sel = order[b0:b0 + B]
            xb, yb = X[sel], y[sel]
            nb = len(sel)
            a, p = forward(params[0], params[1], params[2], params[3][0], xb)
            pc = np.clip(p, feps, 1 - feps)
            loss = -(np.sum(np.where(yb > 0, np.log(pc), 0.0)) + np.sum(np.where(yb < 1, np.log(1 - pc), 0.0))) / nb
            loss += 0.5 * alpha * (np.sum(params[0] ** 2) + np.sum(params[2] ** 2)) / nb
            acc += loss * nb
            delta = p - yb
            gW2 = (a.T @ delta + alpha * params[2]) / nb
            gb2 = np.array([np.mean(delta)])
            dh = np.outer(delta, params[2]) * (1 - a ** 2)
            gW1 = (xb.T @ dh + alpha * params[0]) / nb
            gb1 = np.mean(dh, axis=0)
            t += 1
            lr_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
            for p_, g, m, v in zip(params, [gW1, gb1, gW2, gb2], ms, vs):
                m *= beta1
                m += (1 - beta1) * g
                v *= beta2
                v += (1 - beta2) * g * g
                p_ += -lr_t * m / (np.sqrt(v) + eps)
        curve.append(acc / n)
        if curve[-1] > best - tol:
            no_improve += 1
        else:
            no_improve = 0
        if curve[-1] < best:
            best = curve[-1]
        if no_improve > n_iter_no_change:
            break
    return dict(W1=params[0], b1=params[1], W2=params[2], b2=float(params[3][0]), loss_curve=np.array(curve),
                n_iter=len(curve))


def group_kfold(groups, n_splits=5):
    """scikit-learn GroupKFold (no shuffle): groups sorted by size (descending, stable on the reversed argsort) are dealt
    to the currently lightest fold.  -> fold number per sample."""
    uniq, inv = np.unique(np.asarray(groups), return_inverse=True)
    if len(uniq) < n_splits:
        raise ValueError('Cannot have number of splits n_splits=%d greater than the number of groups: %d.'
                         % (n_splits, len(uniq)))
    per_group = np.bincount(inv)
    indices = np.argsort(per_group)[::-1]
    per_group = per_group[indices]
    per_fold = np.zeros(n_splits)
    group_to_fold = np.zeros(len(uniq), dtype=np.int64)
    for gi, weight in enumerate(per_group):
        lightest = int(np.argmin(per_fold))
        per_fold[lightest] += weight
        group_to_fold[indices[gi]] = lightest
    return group_to_fold[inv]


def accuracy(model, X, y):
    _, p = forward(model['W1'], model['b1'], model['W2'], model['b2'], np.asarray(X, dtype=np.float64))
    return float(np.mean((p > 0.5) == (np.asarray(y) > 0)))
