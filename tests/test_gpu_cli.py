"""GPU: the syncasm CLI itself (run_syncasm.c:342 main, untouched) over the device path.  oracle/_ref/syncasm_dropin is the reference's own
translation units with six hot-path symbols bound to oatk_amd/lib/liboatk_dropin.a and the two consensus hooks (include/oatk_dropin.h;
`make -C oracle ref_dropin`).  Same file, same options as the reference binary: both GFA files byte-identical, and the log must show the MI355X
served every call -- or, for inputs the device path declines, that the ORIGINAL body did and the result is still identical."""
import filecmp
import gzip
import os

import numpy as np
import pytest

import adversarial as A
import cli_util as U
import ref_lib as R

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not U.available(), reason="oracle/_ref CLI binaries not built")]

SIX = ("sr_read", "sr_db_stat", "collect_syncmer_from_reads", "make_syncmer_graph", "read_error_correction", "scg_read_alignment")


def both(tmp_path, files, k, s, c, extra=(), threads=4, env=None):
    ref, dev = str(tmp_path / "ref"), str(tmp_path / "dev")
    U.run_cli(U.CLI_REF, files, ref, k, c, threads, extra=["-s", str(s)] + list(extra))
    e = {"OATK_DROPIN_LOG": "1"}
    e.update(env or {})
    _, err = U.run_cli(U.CLI_DROPIN, files, dev, k, c, threads, env=e, extra=["-s", str(s)] + list(extra))
    for suffix in (".utg.gfa", ".utg.final.gfa"):
        assert os.path.getsize(ref + suffix) > 100, suffix
        assert filecmp.cmp(ref + suffix, dev + suffix, shallow=False), suffix
    return U.served_table(err), err


@pytest.mark.parametrize("K,S,cov,err", [(1001, 31, 8, 0.0008), (301, 21, 6, 0.001)])
def test_cli_fasta_everything_from_the_device(tmp_path, K, S, cov, err):
    reads = A.hifi_like(260, 50000, 9000 if K > 500 else 5000, seed=K + 77, err=err)
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    tab, log = both(tmp_path, fa, K, S, cov)
    for f in SIX:
        assert tab[f][0] >= 1 and tab[f][2] == 0, (f, tab[f], log[-2000:])
    assert tab["make_syncmer_graph"][0] == 2 and tab["sr_db_stat"][0] == 2 and tab["scg_read_alignment"][0] >= 3
    assert tab["scg_syncmer_consensus"][0] > 20 and tab["calc_syncmer_overlap"][0] > 20      # the consensus hooks were served from the device tables


def test_cli_without_ec_and_unzip(tmp_path):
    reads = A.hifi_like(260, 50000, 5000, seed=5)
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    tab, _ = both(tmp_path, fa, 301, 21, 6, extra=["--no-read-ec", "--unzip-round", "0"])
    assert tab["read_error_correction"][0] == 0 and tab["read_error_correction"][2] == 0
    assert tab["sr_read"][0] == 1 and tab["make_syncmer_graph"][0] == 1 and tab["scg_read_alignment"][0] >= 1


def test_cli_two_files_fastq_gz_and_plain(tmp_path):
    reads = A.hifi_like(300, 50000, 5000, seed=9, err=0.0008)
    f1, f2 = str(tmp_path / "a.fq.gz"), str(tmp_path / "b.fq")
    with gzip.open(f1, "wb") as f:
        for i, r in enumerate(reads[:150]):
            f.write(b"@q%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    with open(f2, "wb") as f:
        for i, r in enumerate(reads[150:]):
            f.write(b"@w%d second file\r\n" % i + r + b"\r\n+\r\n" + b"5" * len(r) + b"\r\n")
    tab, _ = both(tmp_path, [f1, f2], 301, 21, 6)
    for f in SIX:
        assert tab[f][0] >= 1 and tab[f][2] == 0, (f, tab[f])


def test_cli_streamed_in_small_windows(tmp_path):
    """the input goes through the device in windows (768 MiB by default); here they are 150 kB, so records straddle window ends all the time"""
    reads = A.hifi_like(300, 50000, 5000, seed=12, err=0.0008)
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    tab, log = both(tmp_path, fa, 301, 21, 6, env={"OATK_DEBUG_WINDOW": "150000"})
    for f in SIX:
        assert tab[f][0] >= 1 and tab[f][2] == 0, (f, tab[f])
    import re
    assert int(re.search(r"of text in (\d+) windows", log).group(1)) >= 8


def test_cli_wrapped_fasta_gz(tmp_path):
    reads = A.hifi_like(300, 50000, 5000, seed=10, err=0.0008)
    fa = str(tmp_path / "w.fa.gz")
    with gzip.open(fa, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">w%d wrapped\n" % i + b"\n".join(r[j:j + 70] for j in range(0, len(r), 70)) + b"\n")
    tab, _ = both(tmp_path, fa, 301, 21, 6)
    for f in SIX:
        assert tab[f][0] >= 1 and tab[f][2] == 0, (f, tab[f])


def test_cli_auto_coverage(tmp_path):
    """-c 0: min_k_cov = 10 x the k-mer peak of sr_db->stats (run_syncasm.c:89-92), i.e. it comes from the device's sr_db_stat.  The heuristic
    is made for an organelle in total-DNA reads: a thin nuclear background (k-mer peak 8) under a 400x plastid-sized genome."""
    nuc = A.hifi_like(700, 400000, 6000, seed=3, err=0.0005)
    org = A.hifi_like(1300, 20000, 6000, seed=4, err=0.0005)
    reads = nuc + org
    reads = [reads[i] for i in np.random.default_rng(5).permutation(len(reads))]
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    tab, log = both(tmp_path, fa, 301, 21, 0)
    assert tab["sr_db_stat"][0] == 2 and tab["sr_read"][0] == 1 and "set minimum kmer coverage as 80" in log
    for f in SIX:
        assert tab[f][2] == 0, (f, tab[f])


def test_cli_inputs_the_reader_used_to_decline(tmp_path):
    """a data cap (-D), wrapped FASTQ, FASTQ and FASTA records in one stream: all read on the device now (OATK_FMT_KSEQ: kseq's own line-by-line reading
    with the classification walked on the host over three numbers per line; the cap by counting the bases of the records as the windows come in) --
    the reference's bytes, nothing from an original body.  What is still declined: k beyond the device scan's window."""
    reads = A.hifi_like(260, 50000, 5000, seed=13, err=0.001)
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    # a data cap stops the reader behind the read that takes the total to the cap (syncmer.c:537-541)
    for cap, win in (("600000", None), ("700k", "150000")):
        env = {"OATK_DEBUG_WINDOW": win} if win else None
        tab, log = both(tmp_path, fa, 301, 21, 6, extra=["-D", cap], env=env)
        assert "data limit" in log
        for f in SIX:
            assert tab[f][0] >= 1 and tab[f][2] == 0, (f, tab[f], log[-2000:])
    # ... and with several handles
    tab, log = both(tmp_path, fa, 301, 21, 6, extra=["-D", "600000"], env={"OATK_DEVICES": "0,0", "OATK_DEBUG_WINDOW": "200000"})
    for f in SIX:
        assert tab[f][0] >= 1 and tab[f][2] == 0, (f, tab[f], log[-2000:])
    # wrapped FASTQ: sequence and quality over several lines, quality lines that start with '@' and '>'
    fq = str(tmp_path / "w.fq")
    with open(fq, "wb") as f:
        for i, r in enumerate(reads):
            h = len(r) // 3
            q = (b"@" + b"I" * (h - 1), b">" + b"5" * (h - 1), b"+" * (len(r) - 2 * h))
            f.write(b"@q%d wrapped\n" % i + r[:h] + b"\n" + r[h:2 * h] + b"\n\n" + r[2 * h:] + b"\n+q%d\n" % i + q[0] + b"\n" + q[1] + b"\n" + q[2] + b"\n")
    for win in (None, "170000"):
        tab, log = both(tmp_path, fq, 301, 21, 6, env={"OATK_DEBUG_WINDOW": win} if win else None)
        for f in SIX:
            assert tab[f][0] >= 1 and tab[f][2] == 0, (f, tab[f], log[-2000:])
    # a FASTQ file followed by a FASTA file, and FASTA first with FASTQ records in the middle: kseq decides per record
    f1, f2, f3 = str(tmp_path / "m.fq"), str(tmp_path / "m.fa"), str(tmp_path / "m2.fa")
    with open(f1, "wb") as f:
        for i, r in enumerate(reads[:130]):
            f.write(b"@q%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    R.write_fasta(reads[130:], f2)
    tab, log = both(tmp_path, [f1, f2], 301, 21, 6, env={"OATK_DEBUG_WINDOW": "300000"})
    for f in SIX:
        assert tab[f][0] >= 1 and tab[f][2] == 0, (f, tab[f], log[-2000:])
    with open(f3, "wb") as f:
        for i, r in enumerate(reads):
            if i % 3 == 1:
                f.write(b"@q%d\r\n" % i + r + b"\r\n+\r\n" + b"I" * len(r) + b"\r\n")
            else:
                f.write(b">a%d\n" % i + r[:2000] + b"\n" + r[2000:] + b"\n")
    tab, log = both(tmp_path, f3, 301, 21, 6)
    for f in SIX:
        assert tab[f][0] >= 1 and tab[f][2] == 0, (f, tab[f], log[-2000:])
    # k beyond the device scan's window (oatk_hip_max_k() = 4016): the original reads and analyses
    long_reads = A.hifi_like(120, 60000, 20000, seed=17, err=0.0)
    fl = str(tmp_path / "long.fa")
    R.write_fasta(long_reads, fl)
    tab, log = both(tmp_path, fl, 4101, 31, 3)
    assert tab["sr_read"][0] == 0 and tab["sr_read"][2] == 1 and "beyond the device" in log


@pytest.mark.parametrize("refuse,originals", [(1, {"read_error_correction": 0, "make_syncmer_graph": 0}),      # EC graph refused: the original BUILDS it, the device still corrects
                                              (2, {"make_syncmer_graph": 1}),                                    # assembly graph refused: original make_syncmer_graph(c, a)
                                              (4, {"collect_syncmer_from_reads": 1, "make_syncmer_graph": 2, "read_error_correction": 1}),
                                              (8, {})])                                                           # distance tables refused: calc_syncmer_overlap walks itself
def test_cli_refusals_of_the_device_fall_back_and_stay_identical(tmp_path, refuse, originals):
    """The shapes the device declines (OATK_E_SPLIT: duplicate (v, w) arcs, an arc with dozens of distances, an oversized hash group) cannot be
    produced from sequence through the scan -- DESIGN.md 10 shows why two consecutive occurrences of one k-mer on one strand always have another
    syncmer between them -- so the refusal itself is requested (OATK_DEBUG_REFUSE) and what is tested is everything behind it: the original
    body runs on the structs the device filled, later calls go on where they still can, and both GFA files stay byte-identical."""
    reads = A.hifi_like(260, 50000, 5000, seed=21 + refuse, err=0.001)
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    ref, dev = str(tmp_path / "ref"), str(tmp_path / "dev")
    U.run_cli(U.CLI_REF, fa, ref, 301, 6, 4, extra=["-s", "21"])
    _, err = U.run_cli(U.CLI_DROPIN, fa, dev, 301, 6, 4, env={"OATK_DROPIN_LOG": "1", "OATK_DEBUG_REFUSE": str(refuse)}, extra=["-s", "21"])
    for suffix in (".utg.gfa", ".utg.final.gfa"):
        assert filecmp.cmp(ref + suffix, dev + suffix, shallow=False), suffix
    tab = U.served_table(err)
    for f, n_orig in originals.items():
        assert tab[f][2] == n_orig, (f, tab[f], err[-1500:])
    assert tab["sr_read"][0] == 1 and tab["sr_read"][2] == 0
    if refuse == 1:
        assert "graph from the original make_syncmer_graph" in err and tab["read_error_correction"][0] == 1
    if refuse == 8:
        assert tab["calc_syncmer_overlap"][0] == 0 and tab["scg_syncmer_consensus"][0] > 20
    if refuse in (1, 2, 8):
        assert tab["scg_read_alignment"][0] >= 3 and tab["scg_read_alignment"][2] == 0


@pytest.mark.parametrize("devices,K,S,cov,err", [("0,0", 1001, 31, 8, 0.0008), ("0,0,0,0", 301, 21, 6, 0.001), ("0,0,0", 101, 11, 5, 0.003)])
def test_cli_over_several_handles(tmp_path, devices, K, S, cov, err):
    """OATK_DEVICES names several handles (here all on the test box's one GPU, talking through the in-process group): the reads are spread over them by
    position in the input, the count tables merged by hash range, the correction sharded, and one syncmer_db_t / one graph / one set of consensus tables
    / one alignment vector handed to the reference's serial tail (include/oatk_multi.h).  Same GFA bytes as the reference, every call
    served from the devices."""
    n = devices.count(",") + 1
    reads = A.hifi_like(320, 50000, 9000 if K > 500 else (5000 if K > 200 else 2500), seed=K + 5 * n, err=err)
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    win = os.path.getsize(fa) // (3 * n) + 1000                          # about three windows per handle, records straddling every boundary
    tab, log = both(tmp_path, fa, K, S, cov, env={"OATK_DEVICES": devices, "OATK_DEBUG_WINDOW": str(win)})
    for f in SIX:
        assert tab[f][0] >= 1 and tab[f][2] == 0, (f, tab[f], log[-3000:])
    assert tab["make_syncmer_graph"][0] == 2 and tab["sr_db_stat"][0] == 2 and tab["scg_read_alignment"][0] >= 3
    assert tab["scg_syncmer_consensus"][0] > 20 and tab["scg_syncmer_consensus"][2] == 0
    assert tab["calc_syncmer_overlap"][0] > 20 and tab["calc_syncmer_overlap"][2] == 0
    import re
    assert int(re.search(r"reads into (\d+) handle", log).group(1)) == n


def test_cli_over_several_handles_two_files_and_an_idle_handle(tmp_path):
    """a short input in one window: everything lands on handle 0 and the others hold no reads -- the collectives still run with empty shards; and the
    same with two files (gzip'ed FASTQ + FASTQ with CRLF) spread over three handles"""
    reads = A.hifi_like(300, 50000, 5000, seed=19, err=0.0008)
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    tab, log = both(tmp_path, fa, 301, 21, 6, env={"OATK_DEVICES": "0,0,0"})
    for f in SIX:
        assert tab[f][0] >= 1 and tab[f][2] == 0, (f, tab[f], log[-3000:])
    f1, f2 = str(tmp_path / "a.fq.gz"), str(tmp_path / "b.fq")
    with gzip.open(f1, "wb") as f:
        for i, r in enumerate(reads[:150]):
            f.write(b"@q%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    with open(f2, "wb") as f:
        for i, r in enumerate(reads[150:]):
            f.write(b"@w%d second file\r\n" % i + r + b"\r\n+\r\n" + b"5" * len(r) + b"\r\n")
    tab, log = both(tmp_path, [f1, f2], 301, 21, 6, env={"OATK_DEVICES": "0,0,0", "OATK_DEBUG_WINDOW": "400000"})
    for f in SIX:
        assert tab[f][0] >= 1 and tab[f][2] == 0, (f, tab[f], log[-3000:])


def test_cli_over_several_handles_without_ec_falls_back_after_the_count(tmp_path):
    """--no-read-ec: the sharded graph is built from corrected chains only, so with several handles the graph and what follows are the original's --
    on the complete sr_db_t / syncmer_db_t the handles left on the host.  Still the reference's bytes."""
    reads = A.hifi_like(260, 50000, 5000, seed=5)
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    tab, _ = both(tmp_path, fa, 301, 21, 6, extra=["--no-read-ec", "--unzip-round", "0"], env={"OATK_DEVICES": "0,0", "OATK_DEBUG_WINDOW": "300000"})
    assert tab["sr_read"][0] == 1 and tab["collect_syncmer_from_reads"][0] == 1 and tab["sr_db_stat"][0] == 1
    assert tab["make_syncmer_graph"][2] == 1


def test_cli_over_several_handles_when_a_host_thread_cannot_start(tmp_path):
    """one host thread per handle runs every collective step (host/multi_host.c run_ranks); when one of them cannot be started the threads that did start
    are sent home before any of them is inside a collective, the call fails as a whole, and the original body serves it: the reference's bytes still,
    and no hang (OATK_DEBUG_FAIL_THREAD is the test hook that makes the start of thread r fail)"""
    reads = A.hifi_like(260, 50000, 5000, seed=23, err=0.0008)
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    tab, log = both(tmp_path, fa, 301, 21, 6, env={"OATK_DEVICES": "0,0,0", "OATK_DEBUG_FAIL_THREAD": "2", "OATK_DEBUG_WINDOW": "300000"})
    assert tab["sr_read"][0] == 1 and tab["sr_read"][2] == 0, tab["sr_read"]                     # (read on one thread: nothing to start)
    assert tab["sr_db_stat"][2] >= 1 and tab["collect_syncmer_from_reads"][2] == 1, (tab["sr_db_stat"], tab["collect_syncmer_from_reads"])
    assert "could not start one host thread per handle" in log, log[-3000:]
