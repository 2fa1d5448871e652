"""`--training_tsv`: labelled `.diffs.<k>.train` rows -> the training dicts.

Same result as the reference's load_mCaller_data.py:3-18 (`tsv2matrix`): per sub-model key and label, the feature rows
and their contexts, in file order.  A label is registered by the first row that carries it (even if that row is left
out); rows with fewer than six features or with a literal `0` feature (an empty slot, written so by
extract_contexts.py:186) are left out."""
from .extract_contexts import base_models


def _usable(fields):
    return len(fields) >= 6 and '0' not in fields


def tsv2matrix(tsvname, base):
    key_of = base_models(base, False)
    signals = {key: {} for key in key_of.values()}
    contexts = {key: {} for key in key_of.values()}
    with open(tsvname, 'r') as rows:
        for row in rows:
            columns = row.split('\t')
            context, features, label = columns[3], columns[4].split(','), columns[6].strip()
            centre = len(context) // 2
            key = key_of[context[centre:centre + 2]]               # KeyError on an unknown pair, like the reference
            by_label = signals[key].setdefault(label, [])
            ctx_by_label = contexts[key].setdefault(label, [])
            if _usable(features):
                by_label.append([float(x) for x in features])
                ctx_by_label.append(context)
    return signals, contexts
