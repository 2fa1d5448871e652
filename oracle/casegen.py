"""Random micro-case generator for the eventalign window machine -- TEST INFRASTRUCTURE ONLY.

Produces small, self-contained inputs (FASTA + nanopolish-eventalign-shaped TSV + FASTQ
[+ positions file]) that exercise the regular path and every quirk listed in SURVEY.md §8(a)
R1-R12: header lines, short lines, unknown contigs, NNNNNN rows, quality-filtered reads, >=8 and
>128 events on one position, skips, multi-M regions, strand flips inside a read, positions going
backwards, the same read name in several blocks, a reverse read starting on a palindromic k-mer,
a site at position 0, contig changes inside and between reads, sites next to N.

Used by tests/golden/make_golden.py (to pin oracle/py_oracle.py against the reference, in the
build container) and by the tests (to compare oracle, C oracle and the HIP path).
"""
import random

PALINDROMES6 = None
_COMP = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A', 'N': 'N'}


def _rc(s):
    return ''.join(_COMP[c] for c in reversed(s))


def _palindromes():
    global PALINDROMES6
    if PALINDROMES6 is None:
        out = []
        for a in 'ACGT':
            for b in 'ACGT':
                for c in 'ACGT':
                    h = a + b + c
                    out.append(h + _rc(h))
        PALINDROMES6 = out
    return PALINDROMES6


def _read_name(rng, style):
    core = '%08x-%04x-%04x' % (rng.getrandbits(32), rng.getrandbits(16), rng.getrandbits(16))
    if style == 0:
        return core + '_Basecall_2D_template'
    if style == 1:
        return core
    if style == 2:
        return core + ':1D_000:template'
    return core + '_Basecall_1D_template'


def _events_per_pos(rng, heavy):
    r = rng.random()
    if heavy and r < 0.03:
        return rng.choice([8, 9, 15, 16, 17, 24, 31, 127, 128, 129, 137, 300])
    if r < 0.50:
        return 1
    if r < 0.75:
        return 2
    if r < 0.87:
        return 3
    if r < 0.93:
        return 4
    if r < 0.97:
        return rng.randint(5, 7)
    return rng.randint(8, 12)


def make_genome(rng, length, motif, with_n):
    seq = [rng.choice('ACGT') for _ in range(length)]
    # plant motif copies, palindromic hexamers and (sometimes) N so the interesting paths occur
    for _ in range(max(1, length // 40)):
        p = rng.randrange(0, max(1, length - len(motif)))
        seq[p:p + len(motif)] = list(motif)
    for _ in range(max(1, length // 60)):
        p = rng.randrange(0, max(1, length - 6))
        seq[p:p + 6] = list(rng.choice(_palindromes()))
    if with_n:
        for _ in range(rng.randint(1, 3)):
            p = rng.randrange(1, length - 1)
            seq[p] = 'N'
            if rng.random() < 0.6:               # put a target base right next to the N
                if rng.random() < 0.5:
                    seq[p - 1] = 'A'
                else:
                    seq[p + 1] = 'T'
    return ''.join(seq)[:length]


def gen_case(seed, flavour=None):
    """Return a dict: fasta, tsv, fastq, positions (or None), args (dict), seed, flavour."""
    rng = random.Random(seed)
    flavours = ['plain', 'plain', 'dense', 'positions', 'train', 'quirk_names', 'quirk_flip',
                'quirk_backwards', 'quirk_pal', 'quirk_pos0', 'qual', 'skips', 'multi_contig',
                'heavy', 'header', 'n_context', 'basec', 'bare_model', 'guppy_model', 'bad_positions']
    if flavour is None:
        flavour = rng.choice(flavours)

    base = 'A'
    motif = rng.choice(['GATC', 'GATC', 'GATC', 'A', 'AT', 'GA', 'CAG', 'TGCA', 'GAATTC'])
    model = 'r95'
    if flavour == 'dense' or (flavour == 'n_context' and rng.random() < 0.7):
        motif = rng.choice(['A', 'AT', 'GA'])
    if flavour == 'basec':
        base, motif, model = 'C', rng.choice(['CG', 'C', 'CCGG']), 'r94'
    if flavour == 'bare_model':
        model = 'r94'
    if flavour == 'guppy_model':
        model = 'CAAY'
    if len(motif) == 1:
        base = motif                      # mCaller.py:159-162

    n_contigs = 2 if flavour == 'multi_contig' or rng.random() < 0.15 else 1
    contigs = []
    for c in range(n_contigs):
        L = rng.randint(70, 420)
        g = make_genome(rng, L, motif if base in motif else 'GATC', flavour == 'n_context')
        if flavour == 'quirk_pos0' and c == 0:
            g = motif + g[len(motif):] if rng.random() < 0.5 else _rc(motif) + g[len(motif):]
            if base not in motif:
                g = base + g[1:]
        contigs.append(('ctg%d' % c if rng.random() < 0.7 else 'ecoli%d' % c, g))

    skip_thresh = 0
    if flavour == 'skips' or rng.random() < 0.15:
        skip_thresh = rng.choice([1, 1, 2])
    qual_thresh = 0
    train = flavour == 'train'

    positions_txt = None
    if flavour in ('positions', 'train', 'bad_positions') or (flavour == 'quirk_pos0' and rng.random() < 0.5):
        lines = []
        comp_base = _COMP[base]
        for cname, g in contigs:
            for p, ch in enumerate(g):
                if ch == base and rng.random() < 0.25:
                    lines.append('%s\t%d\t+\t%s\t' % (cname, p, rng.choice(['m6A', 'A'])))
                if ch == comp_base and rng.random() < 0.25:
                    lines.append('%s\t%d\t-\t%s\t' % (cname, p, rng.choice(['m6A', 'A'])))
            if flavour == 'quirk_pos0':
                if g[0] == base:
                    lines.append('%s\t0\t+\tm6A\t' % cname)
                elif g[0] == comp_base:
                    lines.append('%s\t0\t-\tm6A\t' % cname)
        if flavour == 'bad_positions':
            cname, g = contigs[-1]
            p = rng.randrange(len(g))
            lines.append('%s\t%d\t%s\tm6A\t' % (cname, p, rng.choice('+-')))   # often the wrong base
            if rng.random() < 0.3:
                lines.append('%s\t%d\t+\tm6A\t' % (cname, len(g) + 5))           # beyond the contig
        rng.shuffle(lines)
        positions_txt = '\n'.join(lines) + '\n'
        motif_arg = None
    else:
        motif_arg = motif

    # ---- reads ----
    n_reads = rng.randint(1, 5)
    reads = []
    for r in range(n_reads):
        ci = rng.randrange(n_contigs)
        g = contigs[ci][1]
        maxstart = max(0, len(g) - 6 - 12)
        start = rng.randint(0, maxstart)
        if flavour == 'quirk_pos0' and r == 0:
            start, ci = 0, 0
            g = contigs[0][1]
        npos = rng.randint(8, max(9, len(g) - 6 - start))
        rev = rng.random() < 0.5
        if flavour == 'quirk_pal':
            rev = True
            # start the read on a palindromic hexamer if there is one
            cands = [p for p in range(0, len(g) - 6) if g[p:p + 6] == _rc(g[p:p + 6])]
            if cands:
                start = rng.choice(cands)
                npos = rng.randint(8, max(9, len(g) - 6 - start))
        reads.append(dict(contig=ci, start=start, npos=npos, rev=rev,
                          name=_read_name(rng, rng.randrange(4)), qual=rng.uniform(5.0, 13.0)))
    if flavour == 'quirk_names' and n_reads >= 2:
        # same name in several blocks: adjacent, or separated by another read
        reads[-1]['name'] = reads[0]['name']
        reads[-1]['qual'] = reads[0]['qual']
        if n_reads >= 3 and rng.random() < 0.5:
            reads[1]['name'] = reads[0]['name']
            reads[1]['qual'] = reads[0]['qual']
    if flavour == 'qual':
        qs = sorted(r['qual'] for r in reads)
        qual_thresh = round(qs[len(qs) // 2] + rng.choice([-0.01, 0.01]), 2)

    skip_p = rng.choice([0.0, 0.03, 0.06, 0.12])
    n_p = rng.choice([0.0, 0.04, 0.10])
    rows = []
    if flavour == 'header' or rng.random() < 0.1:
        rows.append('contig\tposition\treference_kmer\tread_name\tstrand\tevent_index\t'
                    'event_level_mean\tevent_stdv\tevent_length\tmodel_kmer\tmodel_mean\t'
                    'model_stdv\tstandardized_level')
    for rd in reads:
        cname, g = contigs[rd['contig']]
        block = []
        for p in range(rd['start'], min(rd['start'] + rd['npos'], len(g) - 5)):
            if rng.random() < skip_p:
                continue
            for _ in range(_events_per_pos(rng, flavour == 'heavy')):
                block.append(p)
                if rng.random() < n_p:
                    block.append(-p - 1)        # an NNNNNN row on the same position
        if flavour == 'quirk_backwards' and len(block) > 6:
            for _ in range(rng.randint(1, 3)):
                i = rng.randrange(0, len(block) - 3)
                j = i + rng.randint(1, 3)
                block[i], block[j] = block[j], block[i]
        n = len(block)
        idx0 = rng.randint(0, 30000)
        flip_at = rng.randrange(1, n) if (flavour == 'quirk_flip' and n > 2) else None
        rev = rd['rev']
        for i, p in enumerate(block):
            if flip_at is not None and i == flip_at:
                rev = not rev
            is_n = p < 0
            if is_n:
                p = -p - 1
            idx = idx0 + n - i if rd['rev'] else idx0 + i
            if flip_at is not None and i >= flip_at and rng.random() < 0.5:
                idx = idx0 + (i if rd['rev'] else n - i)
            ref_kmer = g[p:p + 6]
            model_kmer = 'NNNNNN' if is_n else (_rc(ref_kmer) if rev else ref_kmer)
            mu = 0.0 if is_n else rng.uniform(55.0, 117.0)
            ev = (rng.uniform(60, 120) if is_n else mu + rng.gauss(-0.17, 2.44))
            rows.append('%s\t%d\t%s\t%s\tt\t%d\t%.2f\t%.3f\t%.5f\t%s\t%.2f\t%.2f\t%s' % (
                cname, p, ref_kmer, rd['name'], idx, ev, rng.uniform(0.5, 3.0),
                rng.uniform(0.001, 0.01), model_kmer, mu, 0.0 if is_n else rng.uniform(1, 3),
                'inf' if is_n else '%.2f' % rng.gauss(0, 1)))
            if rng.random() < 0.004:
                rows.append(rng.choice(['', 'garbage line', cname + '\t12\tACGTAC']))
            if rng.random() < 0.004:
                rows.append('nosuchcontig\t%d\t%s\t%s\tt\t%d\t80.00\t1.0\t0.002\t%s\t80.50\t1.5\t0.1' % (
                    p, ref_kmer, rd['name'], idx, ref_kmer))
    tsv = '\n'.join(rows) + '\n'
    while len(tsv) < 700:                      # the reference reads nothing from files < 500 bytes
        tsv += 'padding line with too few tokens\n'

    fasta = ''
    for cname, g in contigs:
        fasta += '>%s some description\n' % cname
        for i in range(0, len(g), 60):
            fasta += g[i:i + 60] + '\n'
    fastq = ''
    seen = set()
    for rd in reads:
        key = rd['name']
        if key in seen:
            continue
        seen.add(key)
        n = rng.randint(30, 80)
        target = rd['qual']
        ph = [max(0, min(40, int(round(rng.gauss(target, 2.0))))) for _ in range(n)]
        fastq += '@%s\n%s\n+\n%s\n' % (rd['name'], ''.join(rng.choice('ACGT') for _ in range(n)),
                                     ''.join(chr(33 + q) for q in ph))
    return dict(seed=seed, flavour=flavour, fasta=fasta, tsv=tsv, fastq=fastq,
                positions=positions_txt,
                args=dict(k=6, skip_thresh=skip_thresh, qual_thresh=qual_thresh, base=base,
                          motif=motif_arg, train=train, model=model))
