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
