"""ctypes loader for the in-tree gfx950 library (oatk_amd/lib/liboatk_hip.so, C ABI in include/oatk_hip.h).

The product path has no CPU fallback: if the library is missing or no MI355X is visible, callers get an
exception, never a silently slower path.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OATK_HIP_LIB") or os.path.join(HERE, "lib", "liboatk_hip.so")   # override: development experiments only
HOST_LIB_PATH = os.path.join(HERE, "lib", "liboatk_host.so")

# names mirror include/oatk_hip.h
OK, E_NODEV, E_ARG, E_STATE, E_SMER, E_SPLIT, E_NOMEM = range(7)
READ_ALIGN = 64
BUF = {name: i for i, name in enumerate([
    "HOCO_L", "N_SCM", "N_NN", "N_LRL", "HO_RL", "HOCO_S", "NN_KEY", "LRL_KEY", "LRL_VAL",
    "SCM_OFF", "POS_MPOS", "POS_SMER", "POS_HASH", "POS_KID",
    "SCM_H", "SCM_S", "SCM_COV", "SCM_OCC_OFF", "SCM_OCC"])}
# include/oatk_hip_ec.h
BUF.update({name: 100 + i for i, name in enumerate([
    "EC_N_SCM", "EC_SCM_OFF", "EC_KMER", "EC_MPOS", "EC_SMER", "EC_SCM_COV", "EC_SCM_DEL", "EC_SCM_OCC_OFF", "EC_SCM_OCC", "EC_ERR_DEL",
    "EC_SCM_FWD", "EC_VTX_SRC", "EC_BLOCK_WORK", "EC_BLOCK_OUT"])})
# include/oatk_hip_ingest.h
BUF.update({name: 160 + i for i, name in enumerate(["INGEST_SEQ", "INGEST_OFF", "INGEST_LEN", "INGEST_HDR"])})
FMT_AUTO, FMT_FASTA, FMT_FASTQ = 0, 1, 2
# include/oatk_hip_cons.h
BUF.update({name: 140 + i for i, name in enumerate(["CONS_SEL", "CONS_SLOT", "CONS_RL", "CONS_MSEQ", "CONS_FIRST", "CONS_TOT"])})
# include/oatk_hip_align.h
BUF.update({name: 200 + i for i, name in enumerate(["RA_ALN_SID", "RA_ALN_OFF", "RA_ALN_S", "RA_FRG_UID", "RA_FRG_UBEG", "RA_FRG_UEND", "RA_FRG_SBEG", "RA_FRG_SEND", "RA_SKIPPED"])})
BUF.update({name: 150 + i for i, name in enumerate(["OVL_KEY", "OVL_OFF", "OVL_DIST", "OVL_CNT", "OVL_TAIL"])})
BUF.update({name: 120 + i for i, name in enumerate([
    "EG_IDX_P", "EG_IDX_N", "EG_ARC_V", "EG_ARC_W", "EG_ARC_LS", "EG_ARC_COV", "EG_ARC_COMP"])})
# include/oatk_hip_graph.h
BUF.update({name: 180 + i for i, name in enumerate([
    "AG_SCM_DEL", "AG_VTX_SCM", "AG_VTX_COV", "AG_IDX_P", "AG_IDX_N", "AG_ARC_V", "AG_ARC_W", "AG_ARC_COV", "AG_ARC_COMP", "AG_ARC_LINK"])})
# include/oatk_hip_multi.h
BUF.update({name: 220 + i for i, name in enumerate(["MG_H", "MG_S", "MG_COV", "MG_L2G", "MG_EC_COV", "MG_EC_DEL", "MG_LCOV"])})
BUF.update({name: 230 + i for i, name in enumerate(["MG_G_H", "MG_G_S", "MG_G_COV", "MG_G_DEL", "MG_G_OCC_OFF", "MG_G_OCC", "MG_POS_GKID"])})
TIMERS = ["hpc", "syncmer", "syncmer_n", "scan_post", "count_place", "count_sort", "count_group", "kmer_hash", "ec_graph", "ec_mark", "ec_solve", "ec_refresh"]

EXPORTS = [
    "oatk_hip_abi_version", "oatk_hip_device_count", "oatk_hip_create", "oatk_hip_destroy", "oatk_hip_last_error",
    "oatk_hip_stream", "oatk_hip_sync", "oatk_hip_max_k", "oatk_hip_scan", "oatk_hip_scan_host", "oatk_hip_count",
    "oatk_comm_unique_id", "oatk_comm_create", "oatk_comm_group_create", "oatk_comm_group_rank", "oatk_comm_group_destroy", "oatk_comm_destroy",
    "oatk_comm_rank", "oatk_comm_size", "oatk_comm_backend", "oatk_comm_traffic", "oatk_hip_merge_counts", "oatk_hip_multi_range", "oatk_hip_ec_sharded",
    "oatk_hip_gather_table", "oatk_hip_asm_graph_sharded", "oatk_hip_consensus_sharded", "oatk_hip_overlap_hist_sharded", "oatk_hip_stat_sharded",
    "oatk_hip_scan_begin", "oatk_hip_scan_reserve", "oatk_hip_scan_append", "oatk_hip_device", "oatk_hip_d2d", "oatk_hip_mem_pool",
    "oatk_hip_info", "oatk_hip_buffer", "oatk_hip_d2h", "oatk_hip_d2h_async", "oatk_hip_h2d_async", "oatk_hip_staging", "oatk_hip_ingest_text_buffer", "oatk_hip_set_timing", "oatk_hip_get_timing",
    "oatk_hip_debug_hash_mask", "oatk_hip_debug_force_general", "oatk_hip_debug_list_cap",
    "oatk_hip_ec_graph", "oatk_hip_ec_graph_light", "oatk_hip_ec", "oatk_hip_ec_stats", "oatk_hip_debug_ec_tiers", "oatk_hip_debug_wf_ed", "oatk_hip_debug_wf_ed_wg", "oatk_hip_ec_mark", "oatk_hip_ec_correct",
    "oatk_hip_ec_set_global", "oatk_hip_ec_pairs", "oatk_hip_ec_graph_from_pairs", "oatk_hip_ec_graph_from_segments", "oatk_hip_ec_export_kmers", "oatk_hip_ec_import_kmers",
    "oatk_hip_ec_reserve_import", "oatk_hip_consensus", "oatk_hip_consensus_ids", "oatk_hip_ingest", "oatk_hip_ingest_host", "oatk_hip_scan_ingested", "oatk_hip_stat", "oatk_hip_stat_keys", "oatk_hip_stat_from_keys",
    "oatk_hip_asm_graph", "oatk_hip_asm_pairs", "oatk_hip_asm_graph_from_pairs", "oatk_hip_overlap_hist", "oatk_hip_overlap_pairs", "oatk_hip_overlap_hist_from_pairs", "oatk_hip_read_alignment", "oatk_hip_debug_align_two_pass",
]


class EcGraph(C.Structure):
    """oatk_ec_graph_t (include/oatk_hip_ec.h): the reference's asmg_t flattened, host pointers"""
    _fields_ = [("n_vtx", C.c_uint64), ("n_arc", C.c_uint64), ("idx_p", C.c_void_p), ("idx_n", C.c_void_p), ("arc_v", C.c_void_p),
                ("arc_w", C.c_void_p), ("arc_ls", C.c_void_p), ("arc_cov", C.c_void_p), ("arc_del", C.c_void_p)]


class RaGraph(C.Structure):
    """oatk_ra_graph_t (include/oatk_hip_align.h): what scg_read_alignment reads from scg_t, flattened, host pointers"""
    _fields_ = [("n_scm", C.c_uint64), ("n_utg", C.c_uint64), ("n_arc", C.c_uint64), ("su_off", C.c_void_p), ("su_uid", C.c_void_p), ("su_pos", C.c_void_p),
                ("utg_n", C.c_void_p), ("idx_p", C.c_void_p), ("idx_n", C.c_void_p), ("arc_w", C.c_void_p), ("arc_ln", C.c_void_p), ("arc_del", C.c_void_p)]


class StatRaw(C.Structure):
    """oatk_stat_raw_t (include/oatk_hip_stat.h)"""
    _fields_ = [("n_reads", C.c_uint64), ("n_syncmers", C.c_uint64), ("sum_dist", C.c_int64), ("n_dist", C.c_uint64),
                ("smer_unique", C.c_uint64), ("kmer_unique", C.c_uint64), ("smer_cnt", C.c_int64 * 1001), ("kmer_cnt", C.c_int64 * 1001),
                ("smer_no_singleton", C.c_int64), ("kmer_no_singleton", C.c_int64)]


class Info(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("seq_bytes", C.c_uint64), ("sid0", C.c_uint64),
                ("k", C.c_int32), ("s", C.c_int32),
                ("n_occ", C.c_uint64), ("n_nn", C.c_uint64), ("n_lrl", C.c_uint64), ("n_scm", C.c_uint64),
                ("scan_retries", C.c_uint32), ("collisions", C.c_uint32)]


class OatkHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load liboatk_hip.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OatkHipError("%s is missing: build it with __graft_entry__.build()" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.oatk_hip_abi_version.restype = C.c_int
    L.oatk_hip_device_count.restype = C.c_int
    L.oatk_hip_create.restype = vp
    L.oatk_hip_create.argtypes = [C.c_int]
    L.oatk_hip_destroy.argtypes = [vp]
    L.oatk_hip_last_error.restype = C.c_char_p
    L.oatk_hip_last_error.argtypes = [vp]
    L.oatk_hip_stream.restype = vp
    L.oatk_hip_stream.argtypes = [vp]
    L.oatk_hip_sync.argtypes = [vp]
    L.oatk_hip_max_k.restype = C.c_int
    L.oatk_hip_scan.argtypes = [vp, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
    L.oatk_hip_scan_host.argtypes = [vp, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
    L.oatk_hip_count.argtypes = [vp]
    L.oatk_comm_unique_id.argtypes = [vp]
    L.oatk_comm_create.restype = vp
    L.oatk_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.oatk_comm_group_create.restype = vp
    L.oatk_comm_group_create.argtypes = [C.c_int]
    L.oatk_comm_group_rank.restype = vp
    L.oatk_comm_group_rank.argtypes = [vp, C.c_int]
    L.oatk_comm_group_destroy.argtypes = [vp]
    L.oatk_comm_destroy.argtypes = [vp]
    L.oatk_comm_rank.argtypes = [vp]
    L.oatk_comm_size.argtypes = [vp]
    L.oatk_comm_backend.restype = C.c_char_p
    L.oatk_comm_backend.argtypes = [vp]
    L.oatk_comm_traffic.restype = None
    L.oatk_comm_traffic.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    L.oatk_hip_merge_counts.argtypes = [vp, vp, C.POINTER(C.c_uint64)]
    L.oatk_hip_multi_range.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.oatk_hip_ec_sharded.argtypes = [vp, vp, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, vp, C.POINTER(C.c_uint64)]
    L.oatk_hip_gather_table.argtypes = [vp, vp, C.c_int]
    L.oatk_hip_asm_graph_sharded.argtypes = [vp, vp, C.c_uint32, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.oatk_hip_consensus_sharded.argtypes = [vp, vp, C.c_uint32]
    L.oatk_hip_overlap_hist_sharded.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.oatk_hip_stat_sharded.argtypes = [vp, vp, C.POINTER(StatRaw)]
    L.oatk_hip_scan_begin.argtypes = [vp, C.c_uint64, C.c_int, C.c_int]
    L.oatk_hip_scan_reserve.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64]
    L.oatk_hip_scan_append.argtypes = [vp, vp]
    L.oatk_hip_device.argtypes = [vp]
    L.oatk_hip_mem_pool.argtypes = [vp, C.c_uint64]
    L.oatk_hip_d2d.argtypes = [vp, vp, vp, C.c_uint64]
    L.oatk_hip_info.argtypes = [vp, C.POINTER(Info)]
    L.oatk_hip_buffer.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.oatk_hip_d2h.argtypes = [vp, vp, vp, C.c_uint64]
    L.oatk_hip_staging.restype = vp
    L.oatk_hip_staging.argtypes = [vp, C.c_uint64]
    L.oatk_hip_set_timing.argtypes = [vp, C.c_int]
    L.oatk_hip_get_timing.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    L.oatk_hip_debug_hash_mask.argtypes = [vp, C.c_uint64]
    L.oatk_hip_debug_force_general.argtypes = [vp, C.c_int]
    L.oatk_hip_debug_list_cap.argtypes = [vp, C.c_int]
    L.oatk_hip_ec_graph.argtypes = [vp]
    L.oatk_hip_ec_graph_light.argtypes = [vp, C.c_uint32]
    L.oatk_hip_ec.argtypes = [vp, C.POINTER(EcGraph), C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double]
    L.oatk_hip_ec_stats.argtypes = [vp, vp]
    L.oatk_hip_debug_ec_tiers.argtypes = [vp, C.c_int, C.c_int]
    L.oatk_hip_debug_wf_ed.argtypes = [vp, C.c_uint64, vp, vp, vp, vp, vp, vp, vp, vp]
    if hasattr(L, "oatk_hip_debug_wf_ed_wg"):      # (an experiments library built before round 5 lacks it)
        L.oatk_hip_debug_wf_ed_wg.argtypes = [vp, C.c_int, C.c_uint64, vp, vp, vp, vp, vp, vp, vp, vp]
    if hasattr(L, "oatk_hip_debug_ed_ab"):      # tools/experiments/liboatk_hip_experiments.so only
        L.oatk_hip_debug_ed_ab.argtypes = [vp, C.c_int, C.c_uint64, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_float)]
    L.oatk_hip_ec_mark.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double]
    L.oatk_hip_ec_correct.argtypes = [vp, C.c_double]
    L.oatk_hip_ec_set_global.argtypes = [vp, C.c_uint64, vp, vp, vp]
    L.oatk_hip_ec_pairs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.oatk_hip_ec_graph_from_pairs.argtypes = [vp, vp, vp, C.c_uint64]
    L.oatk_hip_ec_graph_from_segments.argtypes = [vp, vp, vp, C.c_uint64]
    L.oatk_hip_ec_export_kmers.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32, vp]
    L.oatk_hip_ec_import_kmers.argtypes = [vp, vp, vp, vp, C.c_uint64, C.c_uint32]
    L.oatk_hip_ec_reserve_import.argtypes = [vp, C.c_uint64]
    L.oatk_hip_consensus.argtypes = [vp, C.c_uint32]
    L.oatk_hip_consensus_ids.argtypes = [vp, vp, C.c_uint64]
    L.oatk_hip_ingest.argtypes = [vp, vp, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.oatk_hip_ingest_host.argtypes = [vp, vp, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.oatk_hip_scan_ingested.argtypes = [vp, C.c_uint64, C.c_int, C.c_int]
    L.oatk_hip_stat.argtypes = [vp, C.POINTER(StatRaw)]
    L.oatk_hip_stat_keys.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64), vp]
    L.oatk_hip_stat_from_keys.argtypes = [vp, vp, vp, C.c_uint64, vp, C.POINTER(StatRaw)]
    L.oatk_hip_debug_align_two_pass.argtypes = [vp, C.c_int]
    L.oatk_hip_read_alignment.argtypes = [vp, C.POINTER(RaGraph), vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), vp]
    L.oatk_hip_overlap_pairs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.oatk_hip_overlap_hist_from_pairs.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.oatk_hip_overlap_hist.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.oatk_hip_asm_graph.argtypes = [vp, C.c_uint32, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.oatk_hip_asm_pairs.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.oatk_hip_asm_graph_from_pairs.argtypes = [vp, vp, C.c_uint64, C.c_uint64, vp, vp, C.c_uint32, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    _lib = L
    return L
