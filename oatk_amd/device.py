"""Python host mirror over the C ABI (include/oatk_hip.h): one resident batch of reads on one MI355X.

Vocabulary follows the reference: reads, hoco (homopolymer-compressed) strings, syncmers, occurrences.
`scan` stands for sr_read's per-read analysis (syncmer.c:243-421), `count` for
collect_syncmer_from_reads (syncmer.c:1397-1451).
"""
import os
import ctypes as C

import numpy as np

from . import _lib

_DTYPES = {
    "HOCO_L": np.uint32, "N_SCM": np.uint32, "N_NN": np.uint32, "N_LRL": np.uint32, "HO_RL": np.uint8, "HOCO_S": np.uint8,
    "NN_KEY": np.uint64, "LRL_KEY": np.uint64, "LRL_VAL": np.uint32, "SCM_OFF": np.uint64, "POS_MPOS": np.uint32,
    "POS_SMER": np.uint64, "POS_HASH": np.uint64, "POS_KID": np.uint64, "SCM_H": np.uint64, "SCM_S": np.uint64,
    "SCM_COV": np.uint32, "SCM_OCC_OFF": np.uint64, "SCM_OCC": np.uint64,
    "EC_N_SCM": np.uint32, "EC_SCM_OFF": np.uint64, "EC_KMER": np.uint64, "EC_MPOS": np.uint32, "EC_SMER": np.uint64,
    "EC_SCM_COV": np.uint32, "EC_SCM_DEL": np.uint8, "EC_SCM_OCC_OFF": np.uint64, "EC_SCM_OCC": np.uint64, "EC_ERR_DEL": np.uint8,
    "EC_SCM_FWD": np.uint32, "EC_VTX_SRC": np.uint64, "EC_BLOCK_WORK": np.uint32, "EC_BLOCK_OUT": np.uint32,
    "INGEST_SEQ": np.uint8, "INGEST_OFF": np.uint64, "INGEST_LEN": np.uint32, "INGEST_HDR": np.uint64,
    "CONS_SEL": np.uint32, "CONS_SLOT": np.uint32, "CONS_RL": np.uint32, "CONS_MSEQ": np.uint32, "CONS_FIRST": np.uint64, "CONS_TOT": np.uint64,
    "EG_IDX_P": np.uint64, "EG_IDX_N": np.uint32, "EG_ARC_V": np.uint64, "EG_ARC_W": np.uint64, "EG_ARC_LS": np.uint32,
    "EG_ARC_COV": np.uint32, "EG_ARC_COMP": np.uint8,
    "RA_ALN_SID": np.uint32, "RA_ALN_OFF": np.uint64, "RA_ALN_S": np.float64, "RA_FRG_UID": np.uint64, "RA_FRG_UBEG": np.uint32,
    "RA_FRG_UEND": np.uint32, "RA_FRG_SBEG": np.uint32, "RA_FRG_SEND": np.uint32, "RA_SKIPPED": np.uint32,
    "OVL_KEY": np.uint64, "OVL_OFF": np.uint64, "OVL_DIST": np.int32, "OVL_CNT": np.uint32, "OVL_TAIL": np.uint8,
    "MG_G_H": np.uint64, "MG_G_S": np.uint64, "MG_G_COV": np.uint32, "MG_G_DEL": np.uint8, "MG_G_OCC_OFF": np.uint64, "MG_G_OCC": np.uint64, "MG_POS_GKID": np.uint64,
    "MG_H": np.uint64, "MG_S": np.uint64, "MG_COV": np.uint32, "MG_L2G": np.uint32, "MG_EC_COV": np.uint32, "MG_EC_DEL": np.uint8, "MG_LCOV": np.uint32,
    "AG_SCM_DEL": np.uint8, "AG_VTX_SCM": np.uint32, "AG_VTX_COV": np.uint32, "AG_IDX_P": np.uint64, "AG_IDX_N": np.uint32,
    "AG_ARC_V": np.uint64, "AG_ARC_W": np.uint64, "AG_ARC_COV": np.uint32, "AG_ARC_COMP": np.uint8, "AG_ARC_LINK": np.uint64,
}


def pack_reads(reads):
    """list of bytes -> packed read stream (seq uint8, off uint64[n], len uint32[n]); reads start on 64-byte boundaries"""
    n = len(reads)
    lens = np.fromiter((len(r) for r in reads), dtype=np.uint32, count=n)
    padded = (lens.astype(np.uint64) + (_lib.READ_ALIGN - 1)) // _lib.READ_ALIGN * _lib.READ_ALIGN
    off = np.zeros(n, dtype=np.uint64)
    if n > 1:
        off[1:] = np.cumsum(padded[:-1], dtype=np.uint64)
    total = int(padded.sum()) if n else 0
    seq = np.zeros(max(total, _lib.READ_ALIGN), dtype=np.uint8)
    for i, r in enumerate(reads):
        if len(r):
            seq[int(off[i]):int(off[i]) + len(r)] = np.frombuffer(r, dtype=np.uint8)
    return seq, off, lens


class HipSyncasm:
    """Owns one oatk_hip_ctx.  Raises OatkHipError when the HIP library or the GPU is missing (no fallback)."""

    def __init__(self, device=0):
        self.L = _lib.load()
        self.h = self.L.oatk_hip_create(device)
        if not self.h:
            raise _lib.OatkHipError("oatk_hip_create(%d) failed: no usable MI355X (gfx950) device" % device)
        self.device = device
        if os.environ.get("OATK_TEST_POOL"):        # tests/test_gpu_pool.py: the whole suite of a file over device memory in pieces (include/oatk_hip.h: oatk_hip_mem_pool)
            self.mem_pool(int(os.environ["OATK_TEST_POOL"]))

    def mem_pool(self, warm_bytes=0):
        """device memory in pieces from here on, for every handle of this process on this device (oatk_hip_mem_pool)"""
        self._check(self.L.oatk_hip_mem_pool(self.h, warm_bytes), "oatk_hip_mem_pool")

    def close(self):
        if self.h:
            self.L.oatk_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != _lib.OK:
            msg = self.L.oatk_hip_last_error(self.h)
            raise _lib.OatkHipError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else ""))

    # ---- scan ----
    def scan_host(self, seq, off, lens, k, s, sid0=0):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n = len(off)
        self._keep = (seq, off, lens)
        rc = self.L.oatk_hip_scan_host(self.h, seq.ctypes.data, off.ctypes.data if n else None, lens.ctypes.data if n else None,
                                       n, seq.size if n else 0, sid0, k, s)
        self._check(rc, "oatk_hip_scan_host")

    def scan_device(self, d_seq, d_off, d_len, n_reads, seq_bytes, k, s, sid0=0):
        """device pointers (ints), e.g. torch tensors' data_ptr()"""
        rc = self.L.oatk_hip_scan(self.h, d_seq, d_off, d_len, n_reads, seq_bytes, sid0, k, s)
        self._check(rc, "oatk_hip_scan")

    def scan_begin(self, k, s, sid0=0):
        """start a batch that is assembled from scanned pieces (oatk_hip_scan_begin / _append)"""
        self._check(self.L.oatk_hip_scan_begin(self.h, sid0, k, s), "oatk_hip_scan_begin")

    def scan_reserve(self, seq_bytes, n_reads, n_occ):
        self._check(self.L.oatk_hip_scan_reserve(self.h, seq_bytes, n_reads, n_occ), "oatk_hip_scan_reserve")

    def scan_append(self, piece):
        """move the scan resident in `piece` (another HipSyncasm on this device, scanned with sid0 = the reads held so far) behind this batch"""
        self._check(self.L.oatk_hip_scan_append(self.h, piece.h), "oatk_hip_scan_append")

    def count(self):
        self._check(self.L.oatk_hip_count(self.h), "oatk_hip_count")

    # ---- reads sharded over several GPUs, through the C collectives (include/oatk_hip_multi.h; comm = an oatk_comm pointer) ----
    def merge_counts(self, comm):
        n = C.c_uint64()
        self._check(self.L.oatk_hip_merge_counts(self.h, comm, C.byref(n)), "oatk_hip_merge_counts")
        return int(n.value)

    def multi_range(self):
        """(first global id, syncmers, size of the whole table) of the range of the merged table this handle owns (include/oatk_hip_multi.h)"""
        a, b, g = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.L.oatk_hip_multi_range(self.h, C.byref(a), C.byref(b), C.byref(g)), "oatk_hip_multi_range")
        return int(a.value), int(b.value), int(g.value)

    def ec_sharded(self, comm, max_edist, c, a):
        st = np.zeros(12, np.uint64)
        ni = C.c_uint64()
        self._check(self.L.oatk_hip_ec_sharded(self.h, comm, max_edist, c, 10 * c, c, a, st.ctypes.data, C.byref(ni)), "oatk_hip_ec_sharded")
        return st, int(ni.value)

    # ---- sharded reads up to the graph hand-off (include/oatk_hip_multi.h, second half) ----
    def gather_table(self, comm, root=0):
        """the merged (after merge_counts) or refreshed (after ec_sharded) table with its occurrence lists on rank `root`: fetch MG_G_* there"""
        self._check(self.L.oatk_hip_gather_table(self.h, comm, root), "oatk_hip_gather_table")

    def asm_graph_sharded(self, comm, min_k_cov, min_a_cov_f):
        nv, na = C.c_uint64(), C.c_uint64()
        self._check(self.L.oatk_hip_asm_graph_sharded(self.h, comm, int(min_k_cov), float(min_a_cov_f), C.byref(nv), C.byref(na)), "oatk_hip_asm_graph_sharded")
        return int(nv.value), int(na.value)

    def consensus_sharded(self, comm, min_cov=1):
        self._check(self.L.oatk_hip_consensus_sharded(self.h, comm, int(min_cov)), "oatk_hip_consensus_sharded")

    def overlap_hist_sharded(self, comm, min_cov=0):
        np_, ne = C.c_uint64(), C.c_uint64()
        self._check(self.L.oatk_hip_overlap_hist_sharded(self.h, comm, int(min_cov), C.byref(np_), C.byref(ne)), "oatk_hip_overlap_hist_sharded")
        return int(np_.value), int(ne.value)

    def stat_sharded(self, comm):
        r = _lib.StatRaw()
        self._check(self.L.oatk_hip_stat_sharded(self.h, comm, C.byref(r)), "oatk_hip_stat_sharded")
        return self._stat_dict(r)

    # ---- error correction (include/oatk_hip_ec.h) ----
    def ec_graph(self, light_c=0):
        """make_syncmer_graph(sr_db, scm_db, 0, 0.) + hoco arc overlaps on the device (run_syncasm.c:109-117); with light_c > 0 only what
        read_error_correction(…, err_mer_c = light_c, …, err_arc_c >= err_mer_c, …) needs of it (include/oatk_hip_ec.h: the light graph)"""
        if light_c:
            self._check(self.L.oatk_hip_ec_graph_light(self.h, int(light_c)), "oatk_hip_ec_graph_light")
        else:
            self._check(self.L.oatk_hip_ec_graph(self.h), "oatk_hip_ec_graph")

    def ec(self, max_edist, c, a, graph=None):
        """read_error_correction(sr_db, g, max_edist, c, 10 c, c, a) (run_syncasm.c:124, syncerr.c:819); `graph` = dict of
        host arrays shaped like oatk_ec_graph_t, or None for the graph ec_graph() left resident.  Returns stats[12]."""
        if graph is None:
            rc = self.L.oatk_hip_ec(self.h, None, max_edist, c, 10 * c, c, a)
        else:
            keep = {k: np.ascontiguousarray(graph[k], dtype=dt) for k, dt in
                    (("idx_p", np.uint64), ("idx_n", np.uint64), ("arc_v", np.uint64), ("arc_w", np.uint64), ("arc_ls", np.uint64),
                     ("arc_cov", np.uint32), ("arc_del", np.uint8))}
            g = _lib.EcGraph(int(graph["n_vtx"]), int(graph["n_arc"]), *[keep[k].ctypes.data for k in
                             ("idx_p", "idx_n", "arc_v", "arc_w", "arc_ls", "arc_cov", "arc_del")])
            rc = self.L.oatk_hip_ec(self.h, C.byref(g), max_edist, c, 10 * c, c, a)
        self._check(rc, "oatk_hip_ec")
        st = np.zeros(12, np.uint64)
        self._check(self.L.oatk_hip_ec_stats(self.h, st.ctypes.data), "oatk_hip_ec_stats")
        return st

    def wf_ed(self, jobs, wg=0):
        """the device edit distance on its own (oatk_hip_debug_wf_ed; wg = 1, 2 or 6: through the step of the workgroup solver, oatk_hip_debug_wf_ed_wg): jobs = [(target, query, bw, [ql, ...]), ...] with target / query as
        bytes over ACGT (any case); returns, per job, the list of (score, t_end, q_end) after each query length -- what wf_ed_core
        (levdist.c:265-312, extension mode) leaves in a wf_config_t that is resumed with a longer and longer query"""
        code = np.full(256, 255, np.uint8)
        for i, ch in enumerate(b"ACGT"):
            code[ch] = code[ch + 32] = i
        t_off, q_off, s_off = np.zeros(len(jobs) + 1, np.uint64), np.zeros(len(jobs) + 1, np.uint64), np.zeros(len(jobs) + 1, np.uint64)
        for j, (t, q, _, steps) in enumerate(jobs):
            t_off[j + 1], q_off[j + 1], s_off[j + 1] = t_off[j] + len(t), q_off[j] + len(q), s_off[j] + len(steps)
        tc = code[np.frombuffer(b"".join(j[0] for j in jobs), np.uint8)] if jobs else np.zeros(0, np.uint8)
        qc = code[np.frombuffer(b"".join(j[1] for j in jobs), np.uint8)] if jobs else np.zeros(0, np.uint8)
        if (tc == 255).any() or (qc == 255).any():
            raise ValueError("wf_ed: the device alphabet is the 2-bit one of sr_t.hoco_s (ACGT)")
        bw = np.array([j[2] for j in jobs], np.int32)
        ql = np.array([x for j in jobs for x in j[3]], np.int32)
        out = np.zeros((len(ql), 3), np.int32)
        tc, qc = np.ascontiguousarray(tc), np.ascontiguousarray(qc)
        if wg:
            self._check(self.L.oatk_hip_debug_wf_ed_wg(self.h, wg, len(jobs), tc.ctypes.data, t_off.ctypes.data, qc.ctypes.data, q_off.ctypes.data, bw.ctypes.data,
                                                       ql.ctypes.data, s_off.ctypes.data, out.ctypes.data), "oatk_hip_debug_wf_ed_wg")
        else:
            self._check(self.L.oatk_hip_debug_wf_ed(self.h, len(jobs), tc.ctypes.data, t_off.ctypes.data, qc.ctypes.data, q_off.ctypes.data, bw.ctypes.data,
                                                    ql.ctypes.data, s_off.ctypes.data, out.ctypes.data), "oatk_hip_debug_wf_ed")
        return [[tuple(int(v) for v in out[s]) for s in range(int(s_off[j]), int(s_off[j + 1]))] for j in range(len(jobs))]

    def tables(self, jobs):
        """the two tables of a long arc (oatk_hip_debug_tables; experimental): jobs = [(target, string), ...] over ACGT -> per job (table0, table1), int32[tl + 1] each"""
        code = np.full(256, 255, np.uint8)
        for i, ch in enumerate(b"ACGT"):
            code[ch] = code[ch + 32] = i
        t_off, s_off, o_off = np.zeros(len(jobs) + 1, np.uint64), np.zeros(len(jobs) + 1, np.uint64), np.zeros(len(jobs) + 1, np.uint64)
        for j, (t, q) in enumerate(jobs):
            t_off[j + 1], s_off[j + 1], o_off[j + 1] = t_off[j] + len(t), s_off[j] + len(q), o_off[j] + 2 * (len(t) + 1)
        tc = np.ascontiguousarray(code[np.frombuffer(b"".join(j[0] for j in jobs), np.uint8)])
        sc = np.ascontiguousarray(code[np.frombuffer(b"".join(j[1] for j in jobs), np.uint8)])
        out = np.zeros(int(o_off[-1]), np.int32)
        self.L.oatk_hip_debug_tables.argtypes = [C.c_void_p] + [C.c_uint64] + [C.c_void_p] * 6
        self._check(self.L.oatk_hip_debug_tables(self.h, len(jobs), tc.ctypes.data, t_off.ctypes.data, sc.ctypes.data, s_off.ctypes.data, out.ctypes.data, o_off.ctypes.data), "oatk_hip_debug_tables")
        res = []
        for j, (t, _) in enumerate(jobs):
            a = int(o_off[j])
            res.append((out[a:a + len(t) + 1].copy(), out[a + len(t) + 1:a + 2 * (len(t) + 1)].copy()))
        return res

    def ed_ab(self, pairs, myers):
        """SURVEY 7-5's A/B (oatk_hip_debug_ed_ab): pairs = [(target, query, bw), ...] through the wavefront routine (myers = False) or Myers' bit-vector
        algorithm, one lane per pair (True); returns ([(score, t_end, q_end), ...], kernel milliseconds)"""
        code = np.full(256, 255, np.uint8)
        for i, ch in enumerate(b"ACGT"):
            code[ch] = code[ch + 32] = i
        n = len(pairs)
        t_off, q_off = np.zeros(n + 1, np.uint64), np.zeros(n + 1, np.uint64)
        t_off[1:] = np.cumsum([len(p[0]) for p in pairs])
        q_off[1:] = np.cumsum([len(p[1]) for p in pairs])
        tc = np.ascontiguousarray(code[np.frombuffer(b"".join(p[0] for p in pairs), np.uint8)])
        qc = np.ascontiguousarray(code[np.frombuffer(b"".join(p[1] for p in pairs), np.uint8)])
        if (tc == 255).any() or (qc == 255).any():
            raise ValueError("ed_ab: the device alphabet is ACGT")
        bw = np.array([p[2] for p in pairs], np.int32)
        out = np.zeros((n, 3), np.int32)
        ms = C.c_float()
        self._check(self.L.oatk_hip_debug_ed_ab(self.h, 1 if myers else 0, n, tc.ctypes.data, t_off.ctypes.data, qc.ctypes.data, q_off.ctypes.data, bw.ctypes.data,
                                                out.ctypes.data, C.byref(ms)), "oatk_hip_debug_ed_ab")
        return [tuple(int(v) for v in r) for r in out], float(ms.value)

    def ec_stats(self):
        st = np.zeros(12, np.uint64)
        self._check(self.L.oatk_hip_ec_stats(self.h, st.ctypes.data), "oatk_hip_ec_stats")
        return st

    # the same in steps, and the calls for reads sharded over GPUs (device pointers as ints; oatk_amd/multi.py drives them)
    def ec_mark(self, c, a):
        self._check(self.L.oatk_hip_ec_mark(self.h, c, 10 * c, c, a), "oatk_hip_ec_mark")

    def ec_correct(self, max_edist):
        self._check(self.L.oatk_hip_ec_correct(self.h, max_edist), "oatk_hip_ec_correct")
        return self.ec_stats()

    def ec_set_global(self, n_global, d_l2g, d_cov, d_s):
        self._check(self.L.oatk_hip_ec_set_global(self.h, n_global, d_l2g, d_cov, d_s), "oatk_hip_ec_set_global")

    def ec_pairs(self):
        k, d, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._check(self.L.oatk_hip_ec_pairs(self.h, C.byref(k), C.byref(d), C.byref(n)), "oatk_hip_ec_pairs")
        return k.value, d.value, int(n.value)

    def ec_graph_from_pairs(self, d_keys, d_dist, n):
        self._check(self.L.oatk_hip_ec_graph_from_pairs(self.h, d_keys, d_dist, n), "oatk_hip_ec_graph_from_pairs")

    def ec_export_kmers(self, d_ids, n, d_out, stride, d_rev):
        self._check(self.L.oatk_hip_ec_export_kmers(self.h, d_ids, n, d_out, stride, d_rev), "oatk_hip_ec_export_kmers")

    def ec_import_kmers(self, d_ids, d_rev, d_kmers, n, stride):
        self._check(self.L.oatk_hip_ec_import_kmers(self.h, d_ids, d_rev, d_kmers, n, stride), "oatk_hip_ec_import_kmers")

    # ---- FASTA / FASTQ text -> packed read stream on the device (include/oatk_hip_ingest.h) ----
    def ingest_host(self, text, fmt=0, final=True):
        """text: bytes / uint8 array with the (inflated) content of a FASTA or four-line FASTQ file, or a chunk of one (final=False).
        Returns (n_reads, consumed bytes); the packed stream stays resident: scan_ingested() / fetch('INGEST_*')"""
        t = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray, memoryview)) else np.ascontiguousarray(text, dtype=np.uint8)
        self._keep_text = t
        n, used = C.c_uint64(), C.c_uint64()
        self._check(self.L.oatk_hip_ingest_host(self.h, t.ctypes.data if t.size else None, t.size, fmt, 1 if final else 0, C.byref(n), C.byref(used)),
                    "oatk_hip_ingest_host")
        return int(n.value), int(used.value)

    def ingest_device(self, d_text, n_bytes, fmt=0, final=True):
        n, used = C.c_uint64(), C.c_uint64()
        self._check(self.L.oatk_hip_ingest(self.h, d_text, n_bytes, fmt, 1 if final else 0, C.byref(n), C.byref(used)), "oatk_hip_ingest")
        return int(n.value), int(used.value)

    def scan_ingested(self, k, s, sid0=0):
        self._check(self.L.oatk_hip_scan_ingested(self.h, sid0, k, s), "oatk_hip_scan_ingested")

    # ---- scan statistics (include/oatk_hip_stat.h) ----
    @staticmethod
    def _stat_dict(r):
        return {"n_reads": r.n_reads, "n_syncmers": r.n_syncmers, "sum_dist": r.sum_dist, "n_dist": r.n_dist, "smer_unique": r.smer_unique,
                "kmer_unique": r.kmer_unique, "smer_cnt": np.array(r.smer_cnt, np.int64), "kmer_cnt": np.array(r.kmer_cnt, np.int64),
                "smer_no_singleton": r.smer_no_singleton, "kmer_no_singleton": r.kmer_no_singleton}

    def stat_raw(self):
        """multiplicity histograms of s-mers / k-mers and the distance sum of sr_db_stat (syncmer.c:867), at the batch's current stage"""
        r = _lib.StatRaw()
        self._check(self.L.oatk_hip_stat(self.h, C.byref(r)), "oatk_hip_stat")
        return self._stat_dict(r)

    def stat_keys(self):
        s, k, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        add4 = np.zeros(4, np.int64)
        self._check(self.L.oatk_hip_stat_keys(self.h, C.byref(s), C.byref(k), C.byref(n), add4.ctypes.data), "oatk_hip_stat_keys")
        return s.value, k.value, int(n.value), add4

    def stat_from_keys(self, d_smer, d_kkey, n, add4):
        r = _lib.StatRaw()
        a = np.ascontiguousarray(add4, dtype=np.int64)
        self._check(self.L.oatk_hip_stat_from_keys(self.h, d_smer, d_kkey, n, a.ctypes.data, C.byref(r)), "oatk_hip_stat_from_keys")
        return self._stat_dict(r)

    # ---- base-space consensus (include/oatk_hip_cons.h) ----
    def consensus(self, min_cov=1):
        """rounded mean run lengths of every live syncmer with coverage >= min_cov (scg_syncmer_consensus, syncasm.c:949-1001);
        fetch CONS_SEL / CONS_SLOT / CONS_RL / CONS_MSEQ / CONS_FIRST"""
        self._check(self.L.oatk_hip_consensus(self.h, min_cov), "oatk_hip_consensus")

    def read_alignment(self, graph, old_ra=None):
        """scg_read_alignment (alignment.c:596) of the resident chains against a unitig graph given as a dict of host arrays shaped like
        oatk_ra_graph_t; returns (n_aln, n_frg, stats[3])"""
        dts = (("su_off", np.uint64), ("su_uid", np.uint64), ("su_pos", np.uint32), ("utg_n", np.uint32), ("idx_p", np.uint64), ("idx_n", np.uint64),
               ("arc_w", np.uint64), ("arc_ln", np.uint64), ("arc_del", np.uint8))
        keep = {k: np.ascontiguousarray(graph[k], dtype=dt) for k, dt in dts}
        g = _lib.RaGraph(int(graph["n_scm"]), len(keep["utg_n"]), len(keep["arc_w"]), *[keep[k].ctypes.data for k, _ in dts])
        o = None if old_ra is None else np.ascontiguousarray(old_ra, dtype=np.int64)
        na, nf = C.c_uint64(), C.c_uint64()
        st = np.zeros(3, np.uint64)
        self._check(self.L.oatk_hip_read_alignment(self.h, C.byref(g), None if o is None else o.ctypes.data, C.byref(na), C.byref(nf), st.ctypes.data),
                    "oatk_hip_read_alignment")
        return int(na.value), int(nf.value), st

    def overlap_hist(self):
        """pair-distance tables of every adjacent syncmer pair (calc_syncmer_overlap's tabulation, syncasm.c:477-556); returns (n_pairs, n_entries)"""
        np_, ne = C.c_uint64(), C.c_uint64()
        self._check(self.L.oatk_hip_overlap_hist(self.h, C.byref(np_), C.byref(ne)), "oatk_hip_overlap_hist")
        return int(np_.value), int(ne.value)

    def overlap_pairs(self):
        k, d, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._check(self.L.oatk_hip_overlap_pairs(self.h, C.byref(k), C.byref(d), C.byref(n)), "oatk_hip_overlap_pairs")
        return k.value, d.value, int(n.value)

    def overlap_hist_from_pairs(self, d_keys, d_dist, n):
        np_, ne = C.c_uint64(), C.c_uint64()
        self._check(self.L.oatk_hip_overlap_hist_from_pairs(self.h, d_keys, d_dist, n, C.byref(np_), C.byref(ne)), "oatk_hip_overlap_hist_from_pairs")
        return int(np_.value), int(ne.value)

    def consensus_ids(self, d_ids, n):
        """the same for a device array of ids; with sharded reads the results (CONS_TOT, CONS_MSEQ) are this shard's share"""
        self._check(self.L.oatk_hip_consensus_ids(self.h, d_ids, n), "oatk_hip_consensus_ids")

    def info(self):
        i = _lib.Info()
        self._check(self.L.oatk_hip_info(self.h, C.byref(i)), "oatk_hip_info")
        return {f[0]: getattr(i, f[0]) for f in _lib.Info._fields_}

    def sync(self):
        self._check(self.L.oatk_hip_sync(self.h), "oatk_hip_sync")

    def stream(self):
        return self.L.oatk_hip_stream(self.h)

    def set_timing(self, on=True):
        self.L.oatk_hip_set_timing(self.h, 1 if on else 0)

    def timing(self):
        ms = (C.c_float * len(_lib.TIMERS))()
        self.L.oatk_hip_get_timing(self.h, ms, len(_lib.TIMERS))
        return dict(zip(_lib.TIMERS, [float(x) for x in ms]))

    def debug_list_cap(self, cap):
        self._check(self.L.oatk_hip_debug_list_cap(self.h, cap), "oatk_hip_debug_list_cap")

    def debug_force_general(self, on=True):
        self.L.oatk_hip_debug_force_general(self.h, 1 if on else 0)

    # ---- assembly graph (include/oatk_hip_graph.h) ----
    def asm_graph(self, min_k_cov, min_a_cov_f):
        """make_syncmer_graph(sr_db, scm_db, min_k_cov, min_a_cov_f) + asmg_finalize (run_syncasm.c:138); returns (n_vtx, n_arc)"""
        nv, na = C.c_uint64(), C.c_uint64()
        self._check(self.L.oatk_hip_asm_graph(self.h, min_k_cov, min_a_cov_f, C.byref(nv), C.byref(na)), "oatk_hip_asm_graph")
        return int(nv.value), int(na.value)

    def asm_pairs(self):
        k, n = C.c_void_p(), C.c_uint64()
        self._check(self.L.oatk_hip_asm_pairs(self.h, C.byref(k), C.byref(n)), "oatk_hip_asm_pairs")
        return k.value, int(n.value)

    def asm_graph_from_pairs(self, d_keys, n, n_scm, d_cov, d_del, min_k_cov, min_a_cov_f):
        nv, na = C.c_uint64(), C.c_uint64()
        self._check(self.L.oatk_hip_asm_graph_from_pairs(self.h, d_keys, n, n_scm, d_cov, d_del, min_k_cov, min_a_cov_f, C.byref(nv), C.byref(na)),
                    "oatk_hip_asm_graph_from_pairs")
        return int(nv.value), int(na.value)

    def fetch_asm_graph(self):
        names = ["SCM_DEL", "VTX_SCM", "VTX_COV", "IDX_P", "IDX_N", "ARC_V", "ARC_W", "ARC_COV", "ARC_COMP", "ARC_LINK"]
        return {n.lower(): self.fetch("AG_" + n) for n in names}

    def debug_hash_mask(self, mask):
        self.L.oatk_hip_debug_hash_mask(self.h, C.c_uint64(mask & 0xFFFFFFFFFFFFFFFF))

    # ---- results ----
    def buffer(self, name):
        p = C.c_void_p()
        b = C.c_uint64()
        self._check(self.L.oatk_hip_buffer(self.h, _lib.BUF[name], C.byref(p), C.byref(b)), "oatk_hip_buffer(%s)" % name)
        return p.value, int(b.value)

    def fetch(self, name):
        p, b = self.buffer(name)
        dt = np.dtype(_DTYPES[name])
        out = np.zeros(b // dt.itemsize, dtype=dt)
        if b:
            self._check(self.L.oatk_hip_d2h(self.h, out.ctypes.data, p, b), "oatk_hip_d2h(%s)" % name)
        return out

    def fetch_scan(self, off, with_hash=True):
        """Per-read arrays concatenated in read order -- the flat image of sr_t (syncmer.h:48-70)."""
        off = np.asarray(off, dtype=np.uint64)
        n = len(off)
        hoco_l = self.fetch("HOCO_L")
        n_scm = self.fetch("N_SCM")
        n_nn = self.fetch("N_NN")
        n_lrl = self.fetch("N_LRL")
        ho_rl_slab = self.fetch("HO_RL")
        hoco_s_slab = self.fetch("HOCO_S")
        rl_parts, hs_parts = [], []
        for r in range(n):
            o = int(off[r])
            rl_parts.append(ho_rl_slab[o:o + int(hoco_l[r])])
            hs_parts.append(hoco_s_slab[o // 4:o // 4 + (int(hoco_l[r]) + 3) // 4])
        cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dt)
        out = {
            "hoco_l": hoco_l, "n_scm": n_scm, "n_nn": n_nn, "n_lrl": n_lrl,
            "ho_rl": cat(rl_parts, np.uint8), "hoco_s": cat(hs_parts, np.uint8),
            "n_nucl": (self.fetch("NN_KEY") & np.uint64(0xFFFFFFFF)).astype(np.uint32),
            "ho_l_rl": self.fetch("LRL_VAL"),
            "nn_key": self.fetch("NN_KEY"), "lrl_key": self.fetch("LRL_KEY"),
            "m_pos": self.fetch("POS_MPOS"), "s_mer": self.fetch("POS_SMER"),
        }
        if with_hash:
            out["k_mer"] = self.fetch("POS_HASH")
        return out

    def fetch_count(self):
        """Flat image of syncmer_db_t (syncmer.h:86-114) plus the rewritten per-read k_mer ids."""
        return {
            "n_scm": self.info()["n_scm"],
            "h": self.fetch("SCM_H"), "s": self.fetch("SCM_S"), "cov": self.fetch("SCM_COV"),
            "occ_off": self.fetch("SCM_OCC_OFF"), "occ": self.fetch("SCM_OCC"), "k_id": self.fetch("POS_KID"),
        }
