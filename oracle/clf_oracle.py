"""CPU restatement (numpy) of the two closed-form classifiers the reference's `-c LR` / `-c NBC` hand to `predict_proba` at
extract_contexts.py:199 -- TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.

The arithmetic lives in a third-party dependency that is not under /root/reference: scikit-learn (the reference pins no version;
1.7.2 in the build container).  Restated from its published algorithm:

* LogisticRegression, two classes (train_model.py:55-57): decision = X @ coef_.T + intercept_; predict_proba =
  [1 - expit(decision), expit(decision)]  (sklearn/linear_model/_base.py `_predict_proba_lr`, the one-vs-rest path the liblinear
  solver takes).
* GaussianNB (train_model.py:59-60): joint log likelihood of class c = log(class_prior_[c]) - 0.5 * sum(log(2 pi var_[c]))
  - 0.5 * sum((X - theta_[c])**2 / var_[c]); predict_proba = exp(jll - logsumexp(jll))  (sklearn/naive_bayes.py
  `_joint_log_likelihood`, `predict_log_proba`).

Pinned: tests/golden/models/simple_meta.json holds predict_proba of estimators fitted and run in the build container
(tests/golden/make_golden.py::export_simple_fixture); tests/test_simple_classifiers.py holds this file against them."""
import numpy as np


def logistic_proba(coef, intercept, X):
    d = np.asarray(X, dtype=np.float64) @ np.asarray(coef, dtype=np.float64) + float(intercept)
    out = np.empty_like(d)
    pos = d >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-d[pos]))
    e = np.exp(d[~pos])
    out[~pos] = e / (1.0 + e)
    return out


def gnb_proba(theta, var, prior, X):
    X = np.asarray(X, dtype=np.float64)
    jll = []
    for c in range(2):
        n_ij = -0.5 * np.sum(np.log(2.0 * np.pi * var[c]))
        n_ij = n_ij - 0.5 * np.sum(((X - theta[c]) ** 2) / var[c], axis=1)
        jll.append(np.log(prior[c]) + n_ij)
    jll = np.stack(jll, axis=1)
    mx = jll.max(axis=1)
    lse = mx + np.log(np.exp(jll[:, 0] - mx) + np.exp(jll[:, 1] - mx))
    return np.exp(jll[:, 1] - lse)


def forward(models, X, submodel):
    """models: list of model_io.LogisticWeights / GaussianNBWeights; submodel[i]: which of them scores row i (>= len: NaN)."""
    X = np.asarray(X, dtype=np.float64)
    sub = np.asarray(submodel)
    p = np.full(len(X), np.nan)
    for i, m in enumerate(models):
        sel = sub == i
        if not sel.any():
            continue
        p[sel] = logistic_proba(m.coef, m.intercept, X[sel]) if m.kind == 'logistic' else gnb_proba(m.theta, m.var, m.prior, X[sel])
    return p
