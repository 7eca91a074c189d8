"""GPU: a batch assembled from scanned pieces (oatk_hip_scan_begin / _append: the device-side counterpart of sr_read's batches, syncmer.c:505-533)
must be what ONE scan of all the reads leaves -- every resident array, then the count, the EC graph, the correction and the assembly graph on top."""
import numpy as np
import pytest

import adversarial as A
from oatk_amd import HipSyncasm, pack_reads

pytestmark = pytest.mark.gpu

SCAN = ["HOCO_L", "N_SCM", "N_NN", "N_LRL", "NN_KEY", "LRL_KEY", "LRL_VAL", "SCM_OFF", "POS_MPOS", "POS_SMER", "POS_HASH"]
COUNT = ["POS_KID", "SCM_H", "SCM_S", "SCM_COV", "SCM_OCC_OFF", "SCM_OCC"]
EC = ["EG_ARC_V", "EG_ARC_W", "EG_ARC_LS", "EG_ARC_COV", "EC_N_SCM", "EC_KMER", "EC_MPOS", "EC_SMER", "EC_SCM_COV", "EC_SCM_DEL", "EC_SCM_OCC"]


def per_read(hip, off, hoco_l, name, div):
    slab = hip.fetch(name)
    return np.concatenate([slab[int(o) // div:int(o) // div + (int(l) + div - 1) // div] for o, l in zip(off, hoco_l)]) if len(off) else slab[:0]


@pytest.mark.parametrize("K,S,cuts", [(301, 21, (0, 97, 98, 250, 420)), (1001, 31, (0, 1, 200, 420)), (101, 11, (0, 420))])
def test_assembled_batch_equals_one_scan(hip, K, S, cuts):
    reads = A.hifi_like(400, 60000, 2500 if K < 1000 else 6000, seed=K, err=0.001) + A.reads(K, S, seed=3, scale=0.3)[:20]
    reads = reads[:420]
    reads[7] = b"ACGT" * 50 + b"N" * 7 + reads[7]                       # ambiguous bases and a long homopolymer: the rare-event lists
    reads[300] = reads[300][:500] + b"A" * 700 + reads[300][500:]
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, K, S)
    want = {b: hip.fetch(b) for b in SCAN}
    want["ho_rl"] = per_read(hip, off, want["HOCO_L"], "HO_RL", 1)
    want["hoco_s"] = per_read(hip, off, want["HOCO_L"], "HOCO_S", 4)
    hip.count()
    want.update({b: hip.fetch(b) for b in COUNT})
    hip.ec_graph()
    st_want = hip.ec(0.02, 5, 0.35)
    want.update({b: hip.fetch(b) for b in EC})
    ag_want = (hip.asm_graph(5, 0.35), hip.fetch_asm_graph())

    piece = HipSyncasm(0)
    main = HipSyncasm(0)
    try:
        main.scan_begin(K, S)
        assert main.info()["n_reads"] == 0
        if len(cuts) > 3:
            main.scan_reserve(1 << 20, 100, 1000)                         # too small on purpose: the buffers must grow and keep what they hold
        g_off = []
        base = 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            sq, of, ln = pack_reads(reads[a:b])
            piece.scan_host(sq, of, ln, K, S, sid0=a)
            main.scan_append(piece)
            g_off.extend((of + np.uint64(base)).tolist())
            base += int(sq.size) if b > a else 0
        inf = main.info()
        assert inf["n_reads"] == len(reads) and inf["n_occ"] == len(want["POS_MPOS"])
        got_off = np.array(g_off, np.uint64)
        for b in SCAN:
            assert np.array_equal(main.fetch(b), want[b]), b
        assert np.array_equal(per_read(main, got_off, want["HOCO_L"], "HO_RL", 1), want["ho_rl"])
        assert np.array_equal(per_read(main, got_off, want["HOCO_L"], "HOCO_S", 4), want["hoco_s"])
        main.count()
        for b in COUNT:
            assert np.array_equal(main.fetch(b), want[b]), b
        main.ec_graph()
        st = main.ec(0.02, 5, 0.35)
        assert st[:11].tolist() == st_want[:11].tolist() and int(st[0] + st[5] + st[10]) > 0
        for b in EC:
            assert np.array_equal(main.fetch(b), want[b]), b
        assert main.asm_graph(5, 0.35) == ag_want[0]
        g = main.fetch_asm_graph()
        for k in g:
            assert np.array_equal(g[k], ag_want[1][k]), k
        # the pieces may come from a third handle; an out-of-order piece is refused
        from oatk_amd import OatkHipError
        with pytest.raises(OatkHipError):
            main.scan_append(piece)                                      # its sid0 does not continue the batch
    finally:
        piece.close()
        main.close()
