"""CPU: the arenas of the host mirror (include/oatk_syncasm.h: oatk_host_set_arena and friends).

A program that owns the reference's destroy functions may get the member arrays of many reads as one block.  The registry of blocks has to tell
arena memory from malloc'ed memory wherever a member is freed or replaced: here a database is put together by hand -- members inside an
allocated arena, members inside an adopted block (what the drop-in does with the arrays fetched from the device), members that are blocks of
their own, NULL members, a pointer one past an adopted block -- and cleaned, moved back to blocks of its own, cleaned again.  glibc aborts the
process on a bad free(), so every scenario runs in a child process and must exit cleanly."""
import subprocess
import sys
import textwrap

CHILD = textwrap.dedent(r'''
    import ctypes as C, sys
    sys.path.insert(0, %r)
    from oatk_amd import _lib
    H = C.CDLL(_lib.HOST_LIB_PATH, mode=C.RTLD_GLOBAL)
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]

    class Sr(C.Structure):
        _fields_ = [("sid", C.c_uint64), ("sname", C.c_void_p), ("hoco_l", C.c_uint32), ("hoco_s", C.c_void_p), ("ho_rl", C.c_void_p), ("ho_l_rl", C.c_void_p),
                    ("n_nucl", C.c_void_p), ("n", C.c_uint32), ("m_pos", C.c_void_p), ("s_mer", C.c_void_p), ("k_mer", C.c_void_p)]
    class SrDb(C.Structure):
        _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p), ("k", C.c_int), ("s", C.c_int), ("stats", C.c_void_p)]
    class Scm(C.Structure):
        _fields_ = [("h", C.c_uint64), ("s", C.c_uint64), ("covdel", C.c_uint32), ("m_pos", C.c_void_p)]
    class ScmDb(C.Structure):
        _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p), ("c", C.c_void_p), ("h", C.c_void_p)]

    for f, res, args in (("oatk_host_arena_alloc", C.c_void_p, [C.c_size_t, C.c_void_p]), ("oatk_host_arena_adopt", None, [C.c_void_p, C.c_size_t, C.c_void_p]),
                         ("oatk_sr_member_free", None, [C.c_void_p]), ("oatk_sr_db_clean", None, [C.c_void_p]), ("oatk_sr_db_own_chains", None, [C.c_void_p]),
                         ("oatk_syncmer_db_clean", None, [C.c_void_p]), ("oatk_syncmer_db_own_mpos", None, [C.c_void_p]), ("oatk_host_set_arena", None, [C.c_int]),
                         ("oatk_host_arena", C.c_int, [])):
        getattr(H, f).restype, getattr(H, f).argtypes = res, args

    scenario = sys.argv[1]
    H.oatk_host_set_arena(1)
    assert H.oatk_host_arena() == 1
    N = 300
    db = SrDb()
    db.a = libc.malloc(C.sizeof(Sr) * N)
    db.n = db.m = N
    reads = (Sr * N).from_address(db.a)
    arena = H.oatk_host_arena_alloc(N * 256, C.addressof(db))
    adopted = libc.malloc(N * 64)
    H.oatk_host_arena_adopt(adopted, N * 64, C.addressof(db))
    for i, r in enumerate(reads):
        C.memset(C.addressof(r), 0, C.sizeof(Sr))
        r.sid, r.n, r.hoco_l = i, 4, 100
        r.sname = libc.malloc(16)                                  # names are always blocks of their own
        r.hoco_s, r.ho_rl = arena + i * 256, arena + i * 256 + 32   # a piece's arena
        r.ho_l_rl = libc.malloc(8) if i %% 7 == 0 else None
        if i %% 3 == 0:                                             # chains already replaced by blocks of their own
            r.k_mer, r.m_pos, r.s_mer = libc.malloc(32), libc.malloc(16), libc.malloc(32)
        elif i %% 3 == 1:                                           # chains inside the arena
            r.k_mer, r.m_pos, r.s_mer = arena + i * 256 + 160, arena + i * 256 + 192, arena + i * 256 + 208
        else:                                                      # chains inside an adopted block (the last read's m_pos ends one past it: n = 0 there)
            r.k_mer, r.s_mer = adopted + i * 64, adopted + i * 64 + 32
            r.m_pos = adopted + N * 64 if i == N - 1 else adopted + i * 64 + 16
            if i == N - 1:
                r.n = 0
    sdb = ScmDb()
    sdb.n = sdb.m = 50
    sdb.a = libc.malloc(C.sizeof(Scm) * 50)
    sdb.c, sdb.h = libc.malloc(100), None
    occ = libc.malloc(50 * 24)
    H.oatk_host_arena_adopt(occ, 50 * 24, C.addressof(sdb))
    scms = (Scm * 50).from_address(sdb.a)
    for i, m in enumerate(scms):
        m.h, m.s, m.covdel = i, i, 3
        m.m_pos = None if i %% 5 == 4 else (libc.malloc(24) if i %% 5 == 3 else occ + i * 24)
    if scenario == "own":
        H.oatk_sr_db_own_chains(C.addressof(db))
        H.oatk_syncmer_db_own_mpos(C.addressof(sdb))
        for i, r in enumerate(reads):          # every chain is now a block free() takes
            H.oatk_sr_member_free(r.k_mer); H.oatk_sr_member_free(r.m_pos); H.oatk_sr_member_free(r.s_mer)
            r.k_mer = r.m_pos = r.s_mer = None
    if scenario == "member_free":
        for r in reads:
            H.oatk_sr_member_free(r.hoco_s)    # arena memory: nothing happens
        H.oatk_sr_member_free(None)
    H.oatk_syncmer_db_clean(C.addressof(sdb))
    H.oatk_sr_db_clean(C.addressof(db))
    assert db.n == 0 and not db.a and sdb.n == 0 and not sdb.a
    H.oatk_sr_db_clean(C.addressof(db))        # a clean database cleans again
    print("ok")
''')


def run(scenario):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", CHILD % root, scenario], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0 and p.stdout.strip().endswith(b"ok"), (p.returncode, p.stderr.decode(errors="replace")[-800:])


def test_clean_tells_arena_members_from_blocks_of_their_own():
    run("clean")


def test_members_can_move_back_into_blocks_of_their_own():
    run("own")


def test_member_free_leaves_arena_memory_alone():
    run("member_free")
