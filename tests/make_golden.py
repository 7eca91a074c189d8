#!/usr/bin/env python3
"""Generate tests/golden/*.npz with the COMPILED REFERENCE (oracle/_ref/liboatk_ref.so, built from the
reference's own sources by `make -C oracle ref`).  Runs only in the authoring container, where
/root/reference exists; the fixtures are data (inputs + the reference's outputs), committed so that the
CPU and GPU suites can check the oracle and the HIP path anywhere.

    python tests/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import adversarial as A  # noqa: E402
import ref_lib as R  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def pack(reads):
    off = np.zeros(len(reads) + 1, np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    seq = np.frombuffer(b"".join(reads), dtype=np.uint8)
    return seq, off


def scan_count_case(name, reads, K, S, threads=1):
    n_nn = np.array([sum(1 for c in r if c not in b"ACGTacgtUu\x00\x01\x02\x03") for r in reads], np.uint32)
    db = R.SrDb.from_reads(reads, K, S, threads)
    sc = db.flatten(n_nn=n_nn)
    st_i, st_d = db.stat()
    cdb = R.ScmDb(db)
    cnt = cdb.flatten()
    after = db.flatten()
    seq, off = pack(reads)
    np.savez_compressed(
        os.path.join(GOLD, name + ".npz"), K=K, S=S, seq=seq, off=off,
        hoco_l=sc["hoco_l"], n_scm=sc["n_scm"], n_nn=n_nn, hoco_s=sc["hoco_s"], ho_rl=sc["ho_rl"], ho_l_rl=sc["ho_l_rl"],
        n_nucl=sc["n_nucl"], m_pos=sc["m_pos"], s_mer=sc["s_mer"], k_mer=sc["k_mer"],
        stat_i=st_i, stat_d=st_d,
        scm_h=cnt["h"], scm_s=cnt["s"], scm_cov=cnt["cov"], scm_occ=cnt["occ"], k_id=after["k_mer"])
    print("%-28s K=%-5d S=%-3d reads=%-4d syncmers=%-6d distinct=%d" % (name, K, S, len(reads), len(sc["m_pos"]), cnt["n_scm"]))
    cdb.close()
    db.close()


def levdist_case():
    rng = np.random.default_rng(2024)
    # the reference's own built-in self-test pair (levdist.c:445-446, LEVDIST_TEST_NAIVE prints ED=8 tL=124 t_EN=59 qL=56 q_EN=56)
    ts0 = b"AATGCTCTCATGACATATGAGATAGATACATAGAGACAGATATAGATACACACAGAGATATATGACGTCTGTATGCTCTCTCTCATAGATATACTCTGTAGACTGTCATATACATGCAGAAAAA"
    qs0 = b"CGCTCTCATGACANATGAGATAGATACATAGAGNCAGATATAGATACACACAGTTT"
    assert R.wf_ed(ts0, qs0, -1) == (8, 59, 56)
    T, Q, BW, OUT = [ts0], [qs0], [-1], [R.wf_ed(ts0, qs0, -1)]
    for it in range(600):
        alpha = [b"ACGT", b"AC", b"A"][it % 3]
        ts = A.rand_dna(rng, int(rng.integers(1, 160)), alpha)
        q = bytearray(ts)
        for _ in range(int(rng.integers(0, 7))):
            if not q:
                break
            p, kind = int(rng.integers(0, len(q))), int(rng.integers(0, 3))
            if kind == 0:
                q[p] = alpha[int(rng.integers(0, len(alpha)))]
            elif kind == 1:
                q.insert(p, alpha[int(rng.integers(0, len(alpha)))])
            else:
                del q[p]
        q = bytes(q)
        if it % 4 == 0:
            q = q[:max(1, len(q) // 2)]
        if it % 4 == 1:
            q = q + A.rand_dna(rng, int(rng.integers(1, 60)), alpha)
        q = q or b"A"
        bw = [-1, 2, 6, 12][it % 4]
        w = R.Wavefront(ts, bw)          # wf_ed_core with the caller's initial state (syncerr.c:465-482)
        r = w.step(q)
        w.close()
        T.append(ts), Q.append(q), BW.append(bw), OUT.append(r)
    # resumable traces: one target, a growing query, result after every step (syncerr.c:165-195)
    traces = []
    for it in range(120):
        ts = A.rand_dna(rng, int(rng.integers(10, 400)))
        q = bytearray(ts)
        for _ in range(int(rng.integers(0, 9))):
            p, kind = int(rng.integers(0, len(q))), int(rng.integers(0, 3))
            if kind == 0:
                q[p] = b"ACGT"[int(rng.integers(0, 4))]
            elif kind == 1:
                q.insert(p, b"ACGT"[int(rng.integers(0, 4))])
            else:
                del q[p]
        q = bytes(q) + A.rand_dna(rng, 60)
        bw = int(rng.integers(2, 16))
        w = R.Wavefront(ts, bw)
        ql, steps = 0, []
        while ql < len(q):
            ql = min(len(q), ql + int(rng.integers(1, 140)))
            r = w.step(q[:ql])
            steps.append((ql,) + r)
            if r[0] > bw:
                break
        w.close()
        traces.append((ts, q, bw, steps))
    np.savez_compressed(
        os.path.join(GOLD, "levdist.npz"),
        pairs_t=np.array(T, dtype=object), pairs_q=np.array(Q, dtype=object), pairs_bw=np.array(BW, np.int32),
        pairs_out=np.array(OUT, np.int32),
        tr_t=np.array([t[0] for t in traces], dtype=object), tr_q=np.array([t[1] for t in traces], dtype=object),
        tr_bw=np.array([t[2] for t in traces], np.int32),
        tr_steps=np.array([np.array(t[3], np.int32) for t in traces], dtype=object), allow_pickle=True)
    print("levdist: %d pairs, %d resumable traces" % (len(T), len(traces)))


def ec_case(name, reads, K, S, c, a=0.35, max_edist=0.02):
    """error correction: inputs = the reference's scan/count databases + its EC graph (make_syncmer_graph(…, 0, 0.) and the
    hoco consensus, run_syncasm.c:109-117) flattened; outputs = what read_error_correction (syncerr.c:819) leaves behind."""
    import ec_util as E
    db = R.SrDb.from_reads(reads, K, S, threads=2)
    scm = R.ScmDb(db)
    sr0, sc0 = db.flatten(), scm.flatten()
    g, G = E.ref_graph(db, scm)
    summary = E.reference_ec(db, scm, g, max_edist, c, a)
    sr1, sc1 = db.flatten(), scm.flatten()
    np.savez_compressed(
        os.path.join(GOLD, name + ".npz"), K=K, S=S, c=c, a=a, max_edist=max_edist,
        in_hoco_l=sr0["hoco_l"], in_hoco_s=sr0["hoco_s"], in_n_scm=sr0["n_scm"], in_k_mer=sr0["k_mer"], in_m_pos=sr0["m_pos"], in_s_mer=sr0["s_mer"],
        in_scm_s=sc0["s"], in_scm_cov=sc0["cov"], in_scm_del=sc0["del"],
        g_vtx_len=G["vtx_len"], g_vtx_seq_off=G["vtx_seq_off"], g_seq=G["seq"], g_arc_v=G["arc_v"], g_arc_w=G["arc_w"], g_arc_ls=G["arc_ls"],
        g_arc_cov=G["arc_cov"], g_arc_del=G["arc_del"], g_vtx_del=G["vtx_del"], g_idx_p=G["idx_p"], g_idx_n=G["idx_n"],
        out_n_scm=sr1["n_scm"], out_k_mer=sr1["k_mer"], out_m_pos=sr1["m_pos"], out_s_mer=sr1["s_mer"],
        out_scm_cov=sc1["cov"], out_scm_del=sc1["del"], out_scm_occ=sc1["occ"],
        out_summary=np.array([summary["total"], summary["uncorrected"], summary["corrected"], summary["ambiseq"], summary["ambipath"]], np.int64))
    print("%-28s K=%-5d blocks=%d corrected=%d uncorrected=%d ambiguous=%d/%d" % (
        name, K, summary["total"], summary["corrected"], summary["uncorrected"], summary["ambiseq"], summary["ambipath"]))
    R.lib().refx_scg_destroy(g)
    scm.close()
    db.close()


def flatten_asm(g):
    """make_syncmer_graph(…, c, a) of the reference as arrays: vertices (syncmer, cov), arcs with flags and link ids, index"""
    import ctypes as C
    import ec_util as E
    Gd = E.flatten_graph(g)
    nv, na = Gd["n_vtx"], Gd["n_arc"]
    vn, va0, aln, alink = np.zeros(max(nv, 1), np.uint64), np.zeros(max(nv, 1), np.uint64), np.zeros(max(na, 1), np.uint64), np.zeros(max(na, 1), np.uint64)
    R.lib().refx_graph_flatten2(g, *[C.c_void_p(x.ctypes.data) for x in (vn, va0, aln, alink)])
    assert (vn[:nv] == 1).all() and (aln[:na] == 0).all() and (Gd["arc_ls"][:na] == 0).all() and (Gd["arc_del"][:na] == 0).all() and (Gd["vtx_del"] == 0).all()
    return {"vtx_scm": (va0[:nv] >> np.uint64(1)).astype(np.uint32), "vtx_cov": Gd["vtx_cov"], "arc_v": Gd["arc_v"][:na], "arc_w": Gd["arc_w"][:na],
            "arc_cov": Gd["arc_cov"][:na], "arc_comp": Gd["arc_comp"][:na], "arc_link": alink[:na], "idx_p": Gd["idx_p"], "idx_n": Gd["idx_n"]}


def asmgraph_case(name, reads, K, S, c, a=0.35, max_edist=0.02):
    """the assembly graph, make_syncmer_graph(sr_db, scm_db, c, a) (run_syncasm.c:138): `raw_*` built from the uncorrected databases,
    `ec_*` after the error correction of the golden case `name` (whose out_* arrays are this graph's input chains)"""
    import ec_util as E
    L = R.lib()
    out = {}
    for tag in ("raw", "ec"):
        db = R.SrDb.from_reads(reads, K, S, threads=2)
        scm = R.ScmDb(db)
        if tag == "ec":
            g, _ = E.ref_graph(db, scm)
            E.reference_ec(db, scm, g, max_edist, c, a)
            L.refx_scg_destroy(g)
            gold = np.load(os.path.join(GOLD, name + ".npz"))
            assert np.array_equal(db.flatten()["k_mer"], gold["out_k_mer"])
        g = L.refx_make_graph(db.handle, scm.handle, c, a)
        F = flatten_asm(g)
        for k, v in F.items():
            out[tag + "_" + k] = v
        out[tag + "_scm_del"] = scm.flatten()["del"]
        print("%-28s %-3s vertices=%d arcs=%d dropped syncmers=%d" % ("asmgraph_" + name, tag, len(F["vtx_scm"]), len(F["arc_v"]), int(out[tag + "_scm_del"].sum())))
        L.refx_scg_destroy(g)
        scm.close()
        db.close()
    np.savez_compressed(os.path.join(GOLD, "asmgraph_" + name + ".npz"), c=c, a=a, **out)


def asmgraph_cases():
    import test_gpu_ec as T
    asmgraph_case("ec_diploid_k101", T.diploid_reads(101, 6000, 150, 500, 1200, 0.006), 101, 11, 4)
    asmgraph_case("ec_repeats_k301", T.sample_reads(T.genome_with_repeats(7, 25000, unit=1500, copies=4), 320, 4000, 0.003, 8), 301, 21, 5)
    asmgraph_case("ec_hifi_k1001", A.hifi_like(120, 30000, 9000, seed=1009, err=0.0008), 1001, 31, 6)


def align_case(name, reads, K, S, c):
    """read -> unitig alignment (scg_read_alignment, alignment.c:596): the corrected chains, and per stage of the reference's pipeline
    (one syncmer per vertex, after unitigging, unzip rounds with multiplexing, final) the flattened graph, the old_ra filter and the
    alignments the compiled reference produced"""
    import test_oracle_align as TA
    import align_util as AU
    chains, stages = TA.reference_stages(reads, K, S, c)
    out = {"K": K, "S": S, "c": c, "n_scm": chains[0], "k_mer": chains[1], "m_pos": chains[2], "n_stages": len(stages),
           "n_scm_table": stages[0][1]["n_scm"], "stage_names": np.array([st[0] for st in stages])}
    for i, (nm, graph, old, want) in enumerate(stages):
        for k, _ in AU.GRAPH_FIELDS:
            out["s%d_%s" % (i, k)] = graph[k]
        out["s%d_old_ra" % i] = old
        for k in AU.OUT_FIELDS:
            out["s%d_out_%s" % (i, k)] = want[k]
        print("%-28s %-9s unitigs=%d arcs=%d alignments=%d fragments=%d" % ("align_" + name, nm, len(graph["utg_n"]), len(graph["arc_w"]), len(want["sid"]), len(want["uid"])))
    np.savez_compressed(os.path.join(GOLD, "align_" + name + ".npz"), **out)


def align_cases():
    import test_gpu_ec as T
    align_case("repeats_k301", T.sample_reads(T.genome_with_repeats(7, 25000, unit=1500, copies=4), 320, 4000, 0.003, 8), 301, 21, 5)
    align_case("diploid_k101", T.diploid_reads(101, 6000, 150, 500, 1200, 0.006), 101, 11, 4)


def topties_cases():
    for K in (1001, 991):                     # K - S = 970 (window start off the chunk grid) and 960 (on it)
        scan_count_case("topties_k%d_s31" % K, A.top_word_tie_reads(K), K, 31, threads=2)


def main():
    if not R.available():
        sys.exit("oracle/_ref/liboatk_ref.so missing: run `make -C oracle ref` (needs /root/reference)")
    os.makedirs(GOLD, exist_ok=True)
    if "--asmgraph-only" in sys.argv:        # adds asmgraph_*.npz next to existing ec_*.npz without rewriting those
        return asmgraph_cases()
    if "--align-only" in sys.argv:
        return align_cases()
    if "--topties-only" in sys.argv:         # r03h: s-mer pairs whose hashes tie on the top word only (adversarial.top_word_tie_reads)
        return topties_cases()
    for (K, S) in [(101, 11), (61, 15), (33, 31), (25, 5), (64, 16)]:
        scan_count_case("adversarial_k%d_s%d" % (K, S), A.reads(K, S, scale=0.5), K, S)
    scan_count_case("adversarial_k1001_s31", A.reads(1001, 31, scale=0.5), 1001, 31)
    scan_count_case("hifi_k1001_s31", A.hifi_like(120, 60000, 9000, seed=21), 1001, 31, threads=3)
    scan_count_case("hifi_k101_s11", A.hifi_like(200, 8000, 1500, seed=22) + A.reads(101, 11, seed=9, scale=0.3), 101, 11)
    levdist_case()
    import test_gpu_ec as T     # read generators shared with the GPU suite
    ec_case("ec_diploid_k101", T.diploid_reads(101, 6000, 150, 500, 1200, 0.006), 101, 11, 4)
    ec_case("ec_repeats_k301", T.sample_reads(T.genome_with_repeats(7, 25000, unit=1500, copies=4), 320, 4000, 0.003, 8), 301, 21, 5)
    ec_case("ec_hifi_k1001", A.hifi_like(120, 30000, 9000, seed=1009, err=0.0008), 1001, 31, 6)
    asmgraph_cases()
    align_cases()
    topties_cases()


if __name__ == "__main__":
    main()
