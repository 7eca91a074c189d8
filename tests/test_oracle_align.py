"""CPU: the oracle's read alignment (oracle/align.c) against golden vectors made by the compiled reference's scg_read_alignment
(alignment.c:596), and side by side with it through unitigging and unzip rounds (run_syncasm.c:160-232)."""
import ctypes as C
import math

import numpy as np
import pytest

import align_util as AU
import golden_util as G
import ref_lib as R

CASES = ["align_repeats_k301", "align_diploid_k101"]


@pytest.mark.parametrize("case", CASES)
def test_oracle_alignment_matches_reference_golden(case):
    g = G.load(case)
    most = 0
    for st in range(int(g["n_stages"])):
        pre = "s%d_" % st
        graph = {k: g[pre + k] for k, _ in AU.GRAPH_FIELDS}
        graph["n_scm"] = int(g["n_scm_table"])
        got = AU.oracle_align(g["n_scm"], g["k_mer"], g["m_pos"], graph, g[pre + "old_ra"])
        AU.assert_same(got, {k: g[pre + "out_" + k] for k in AU.OUT_FIELDS}, (case, st))
        most = max(most, len(got["sid"]))
    assert most > 100 and int(g["n_stages"]) >= 5


def ref_flatten(L, v):
    na, nf = C.c_uint64(), C.c_uint64()
    L.refx_ra_dims.argtypes = [C.c_void_p] * 3
    L.refx_ra_flatten.argtypes = [C.c_void_p] * 9
    L.refx_ra_dims(v, C.byref(na), C.byref(nf))
    na, nf = na.value, nf.value
    out = {"sid": np.zeros(na, np.uint64), "n": np.zeros(na, np.uint32), "s": np.zeros(na, np.float64), "uid": np.zeros(nf, np.uint64),
           "u_beg": np.zeros(nf, np.uint64), "u_end": np.zeros(nf, np.uint64), "s_beg": np.zeros(nf, np.uint32), "s_end": np.zeros(nf, np.uint32)}
    L.refx_ra_flatten(v, *[out[k].ctypes.data for k in AU.OUT_FIELDS])
    return out


def reference_stages(reads, K, S, c, max_rounds=3):
    """the reference's pipeline up to the unzip rounds; yields (stage name, flattened graph, old_ra, reference alignments, chains)"""
    import ec_util as E
    L = R.lib()
    L.refx_ra_new.restype = C.c_void_p
    L.refx_ra_destroy.argtypes = [C.c_void_p]
    L.refx_read_alignment.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.refx_process_unitigs.argtypes = [C.c_void_p]
    L.refx_update_utg_cov.argtypes = [C.c_void_p]
    L.refx_multiplex.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.c_double]
    db = R.SrDb.from_reads(reads, K, S, threads=2)
    scm = R.ScmDb(db)
    g, _ = E.ref_graph(db, scm)
    E.reference_ec(db, scm, g, 0.02, c, 0.35)
    L.refx_scg_destroy(g)
    sr = db.flatten()
    chains = (sr["n_scm"], sr["k_mer"], sr["m_pos"])
    g = L.refx_make_graph(db.handle, scm.handle, c, 0.35)
    v = L.refx_ra_new()
    stages = []

    def call(name, for_unzip):
        old = AU.old_ra_filter(ref_flatten(L, v), len(sr["n_scm"]), for_unzip)
        graph = AU.ref_ra_graph(g)
        L.refx_read_alignment(db.handle, v, g, 3, for_unzip)
        stages.append((name, graph, old, ref_flatten(L, v)))

    call("vertices", 0)
    L.refx_process_unitigs(g)
    call("unitigs", 0)
    for i in range(max_rounds):
        call("unzip%d" % i, 1)
        L.refx_update_utg_cov(g)
        if L.refx_multiplex(g, v, int(math.ceil(30000.0 / K)), 10.0, 0.3) == 0:
            break
    call("after", 1)
    call("final", 0)
    L.refx_ra_destroy(v)
    L.refx_scg_destroy(g)
    scm.close()
    db.close()
    return chains, stages


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("K,S,c,seed,err", [(101, 11, 4, 3, 0.004), (301, 21, 5, 4, 0.002), (1001, 31, 6, 5, 0.0008)])
def test_oracle_alignment_side_by_side(K, S, c, seed, err):
    import adversarial as A
    import test_gpu_ec as T
    reads = A.hifi_like(240, 30 * K, 6 * K, seed=seed, err=err) + T.sample_reads(T.genome_with_repeats(seed, 20 * K, unit=3 * K, copies=3), 120, 5 * K, err, seed + 1)
    chains, stages = reference_stages(reads, K, S, c)
    assert len(stages) >= 5
    for name, graph, old, want in stages:
        got = AU.oracle_align(*chains, graph, old)
        AU.assert_same(got, want, name)
    assert any(len(w["sid"]) > 20 for _, _, _, w in stages)


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(4))
def test_oracle_alignment_on_random_graphs_side_by_side(seed):
    """graphs no assembler would build (tests/align_util.py: random_graph), handed to the compiled reference's scg_read_alignment through a
    hand-made scg_t: ties, several best chains per read, noise arcs, random old_ra filters -- the oracle reproduces every alignment"""
    import adversarial as A
    import ec_util as E
    L = R.lib()
    L.refx_scg_from_flat.restype = C.c_void_p
    L.refx_scg_from_flat.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64] + [C.c_void_p] * 10
    L.refx_scg_flat_destroy.argtypes = [C.c_void_p]
    L.refx_ra_new.restype = C.c_void_p
    L.refx_ra_destroy.argtypes = [C.c_void_p]
    L.refx_ra_build.restype = C.c_void_p
    L.refx_ra_build.argtypes = [C.c_uint64] + [C.c_void_p] * 8
    L.refx_read_alignment.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    K, S, c = 101, 11, 3
    db = R.SrDb.from_reads(A.hifi_like(300, 12000, 2500, seed=70 + seed, err=0.002), K, S, threads=2)
    scm = R.ScmDb(db)
    g0, _ = E.ref_graph(db, scm)
    E.reference_ec(db, scm, g0, 0.02, c, 0.35)
    L.refx_scg_destroy(g0)
    sr = db.flatten()
    chains = (sr["n_scm"], sr["k_mer"], sr["m_pos"])
    n_table = scm.flatten()["n_scm"]
    rng = np.random.default_rng(seed)
    total = multi = 0
    for n_src, max_len, p_overlap, p_noise in ((25, 1, 0.0, 0.2), (25, 4, 0.3, 0.3), (60, 12, 0.5, 0.5), (12, 2, 0.5, 2.0)):
        G = AU.random_graph(rng, n_table, chains, n_src, max_len, p_overlap, p_noise)
        keep = [np.ascontiguousarray(G[k], dtype=dt) for k, dt in AU.GRAPH_FIELDS]
        arc_v = np.ascontiguousarray(G["arc_v"], dtype=np.uint64)
        order = [keep[0], keep[1], keep[2], keep[3], keep[4], keep[5], arc_v, keep[6], keep[7], keep[8]]
        g = L.refx_scg_from_flat(scm.handle, len(G["utg_n"]), len(G["arc_w"]), *[a.ctypes.data for a in order])
        # first call: every read; second: the for_unzip filter the reference derives from the first call's alignments
        v = L.refx_ra_new()
        L.refx_read_alignment(db.handle, v, g, 3, 0)
        want = ref_flatten(L, v)
        AU.assert_same(AU.oracle_align(*chains, G), want, (seed, n_src, "all"))
        old = AU.old_ra_filter(want, len(chains[0]), 1)
        L.refx_read_alignment(db.handle, v, g, 3, 1)
        want1 = ref_flatten(L, v)
        AU.assert_same(AU.oracle_align(*chains, G, old), want1, (seed, n_src, "unzip"))
        total += len(want["sid"]) + len(want1["sid"])
        multi += int((np.bincount(want["sid"].astype(np.int64)) > 1).sum()) if len(want["sid"]) else 0
        L.refx_ra_destroy(v)
        L.refx_scg_flat_destroy(g)
    assert total > 100 and multi > 0
    scm.close()
    db.close()
