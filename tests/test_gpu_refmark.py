"""The site masks made on the GPU (mc_ctx_set_reference_motif) against the host's marking (refmark.py, the literal statement of
extract_contexts.py:33-81): bases, both strand masks, the site numbering, for every motif the device accepts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    from mcaller_amd.device import Device
    return Device(0)


def test_masks_from_the_device_equal_the_hosts_marking(dev, tmp_path):
    from mcaller_amd import refmark
    rng = np.random.default_rng(11)
    seqs = {'big': ''.join(rng.choice(list('ACGT'), 300001)), 'tiny': 'GATC', 'empty_like': 'A',
            'with_m_and_n': 'ACGMTNNNGATCMMGATCGATC' * 40, 'mid': ''.join(rng.choice(list('ACGTN'), 5000))}
    seqs['big'] = seqs['big'][:700].lower() + seqs['big'][700:150000] + 'gatcGATCgAtC' + seqs['big'][150000:] + 'GATC'
    fa = str(tmp_path / 'r.fa')
    with open(fa, 'w') as fh:
        for name, seq in seqs.items():
            fh.write('>%s extra words\n' % name + '\n'.join(seq[i:i + 60] for i in range(0, len(seq), 60)) + '\n')
    accepted = 0
    for motif, base in (('GATC', 'A'), ('A', 'A'), ('C', 'C'), ('GATC', 'C'), ('CCAGG', 'C'), ('GAT', 'A'), ('AC', 'A'), ('ACGT', 'C'),
                        ('AAAA', 'A'), ('GATCGA', 'A'), ('GG', 'A'), ('GANTC', 'A')):
        ref = refmark.MarkedReference(fa, base, motif, None)
        ref.quiet = True
        dm = ref.motif_for_the_device()
        bordered = any(m[:i] == m[-i:] for m in (motif, refmark.revcomp(motif)) for i in range(1, len(m)))
        assert (dm is None) == bordered, (motif, base)
        if dm is None:
            continue
        accepted += 1
        raw = ref.raw_arrays()
        dev.set_reference_motif(raw, *dm)
        n_contigs = len(ref.records)
        seq, mf, mr, rf, rr, site_base, n_sites = dev.fetch_reference(int(raw['n_seq_bytes']), int(raw['n_words']), n_contigs)
        for cid in range(n_contigs):
            ref.mark(cid)
        want = ref.device_arrays()                      # every contig marked: the same layout
        assert np.array_equal(raw['contig_len'], want['contig_len']) and np.array_equal(raw['word_off'], want['word_off'])
        assert np.array_equal(raw['seq_off'], want['seq_off'])
        assert np.array_equal(seq, want['seq'][:len(seq)]), (motif, base)
        assert np.array_equal(mf, want['mbits_fwd']) and np.array_equal(mr, want['mbits_rev']), (motif, base)
        # the site numbering: marked sites before every word, per contig and strand; (contig, strand) bases in order
        run, k = 0, 0
        for cid in range(n_contigs):
            w0 = int(want['word_off'][cid])
            w1 = int(want['word_off'][cid + 1]) if cid + 1 < n_contigs else len(want['mbits_fwd'])
            for bits, rank in ((want['mbits_fwd'], rf), (want['mbits_rev'], rr)):
                pc = np.array([bin(int(x)).count('1') for x in bits[w0:w1]], dtype=np.int64)
                assert np.array_equal(rank[w0:w1], np.concatenate([[0], np.cumsum(pc)[:-1]])), (motif, base, cid)
                assert site_base[k] == run
                run += int(pc.sum())
                k += 1
        assert n_sites == run
    assert accepted >= 8
