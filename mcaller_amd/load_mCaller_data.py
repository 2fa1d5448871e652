"""`--training_tsv`: labelled `.diffs.<k>.train` rows -> the training dicts (load_mCaller_data.py:3-18)."""
from .extract_contexts import base_models


def tsv2matrix(tsvname, base):
    base_model = base_models(base, False)
    signals, contexts = {bm: {} for bm in base_model.values()}, {bm: {} for bm in base_model.values()}
    with open(tsvname, 'r') as infi:
        for line in infi:
            context, sigs, strand, label = line.split('\t')[3:7]
            label = label.strip()
            centre = int(len(context) / 2)
            twobase_model = base_model[context[centre:centre + 2]]
            if label not in signals[twobase_model]:
                signals[twobase_model][label] = []
                contexts[twobase_model][label] = []
            if len(sigs.split(',')) >= 6 and len([x for x in sigs.split(',') if x == '0']) == 0:     # rows with skips are left out
                signals[twobase_model][label].append([float(s) for s in sigs.split(',')])
                contexts[twobase_model][label].append(context)
    return signals, contexts
