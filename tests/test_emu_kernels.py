"""CPU: the real kernel sources (zarc_amd/csrc/*.hip) compiled against the HIP emulator and driven through the
same C ABI, checked against the oracle.  Catches logic errors, divergent barriers and (with the asan target)
out-of-bounds accesses before any GPU time is spent.  Small inputs: the emulator is slow."""
import parity_cases as pc
from zarc_amd import _lib


def test_emu_blake3(emu_engine, oracle, corpus):
    pc.check_blake3(emu_engine, oracle, corpus, big=False)


def test_emu_xxh64(emu_engine, oracle, corpus):
    pc.check_xxh64_device(emu_engine, oracle, corpus, big=False)


def test_emu_pack_bit_exact_and_valid(emu_engine, oracle, corpus, libzstds):
    pc.check_pack(emu_engine, oracle, corpus, libzstds, big=False)


def test_emu_level_tiers(emu_engine, oracle, corpus, libzstds):
    pc.check_levels(emu_engine, oracle, corpus, libzstds, big=False)


def test_emu_unpack_libzstd_golden(emu_engine, oracle, corpus, golden_frames):
    pc.check_unpack_golden(emu_engine, oracle, corpus, golden_frames, limit=140000)


def test_emu_unpack_error_statuses(emu_engine, oracle, corpus, golden_frames):
    pc.check_unpack_errors(emu_engine, oracle, corpus, golden_frames)


def test_emu_params(emu_engine):
    pc.check_params(emu_engine)


def test_emu_store_mode(emu_engine, oracle, corpus, libzstds):
    pc.check_store(emu_engine, oracle, corpus, libzstds)


def test_emu_unpack_fuzz_agrees_with_oracle(emu_engine, oracle, corpus, golden_frames):
    ok, bad = pc.check_unpack_fuzz(emu_engine, oracle, corpus, golden_frames, n_mut=160, seed=1, max_raw=70000)
    assert bad > 40 and ok >= 0


def test_emu_host_staging_in_chunks(emu_engine, oracle, corpus, golden_frames):
    """The host-pointer entry points cut a batch into chunks and overlap their copies: with a tiny chunk size every
    batch below runs through many chunks, double-buffered arenas and the helper thread; results must not change."""
    emu_engine.set_parameter(_lib.PX_STAGE_CHUNK, 20000)
    try:
        pc.check_roundtrip(emu_engine, oracle, corpus, big=False)
        pc.check_unpack_errors(emu_engine, oracle, corpus, golden_frames)
        pc.check_store(emu_engine, oracle, corpus, [])
    finally:
        emu_engine.set_parameter(_lib.PX_STAGE_CHUNK, 0)


def test_emu_unpack_in_size_groups(emu_engine, oracle, corpus, golden_frames):
    """Unpack deals the frames by descending size into groups whose stages overlap on their own streams (engine.hip,
    zarc_gpu_unpack_batch_device); by itself it does so only for batches with large frames.  Forced here on small ones: golden
    frames, error statuses (the inline decoder runs per group), fuzzed frames and round trips must not change, and results must
    come back in the caller's order."""
    for g in (2, 3):
        emu_engine.set_parameter(_lib.PX_DEC_GROUPS, g)
        try:
            pc.check_unpack_golden(emu_engine, oracle, corpus, golden_frames, limit=70000)
            pc.check_unpack_errors(emu_engine, oracle, corpus, golden_frames)
            pc.check_roundtrip(emu_engine, oracle, corpus, big=False)
            if g == 3:
                ok, bad = pc.check_unpack_fuzz(emu_engine, oracle, corpus, golden_frames, n_mut=60, seed=3, max_raw=70000)
                assert bad > 10
        finally:
            emu_engine.set_parameter(_lib.PX_DEC_GROUPS, 0)
    # fewer frames than groups
    emu_engine.set_parameter(_lib.PX_DEC_GROUPS, 4)
    try:
        raw = corpus.entry(77, 50000, 0)
        (frame, dig), = emu_engine.pack([raw])
        (out, d2, st), = emu_engine.unpack([frame], [len(raw)], [dig])
        assert st == _lib.FRAME_OK and out == raw and d2 == dig
    finally:
        emu_engine.set_parameter(_lib.PX_DEC_GROUPS, 0)


def test_emu_pack_in_sub_batches(emu_engine, oracle, corpus, libzstds):
    """A batch whose encoder scratch does not fit the budget is packed in several sub-batches (scratch reused between them):
    frames must not change."""
    emu_engine.set_parameter(_lib.PX_SCRATCH_MB, 1)
    try:
        pc.check_pack(emu_engine, oracle, corpus, libzstds, big=False)
    finally:
        emu_engine.set_parameter(_lib.PX_SCRATCH_MB, 0)


def test_emu_unpack_in_sub_batches(emu_engine, oracle, corpus, golden_frames):
    """A batch whose decoder scratch (sequences / literals decoded ahead, tables) exceeds the budget is unpacked in halves that may
    split again (engine.hip, unpack_device_split): same bytes, digests, statuses, in the caller's order."""
    emu_engine.set_parameter(_lib.PX_SCRATCH_MB, 1)
    try:
        pc.check_unpack_golden(emu_engine, oracle, corpus, golden_frames, limit=140000)
        pc.check_unpack_errors(emu_engine, oracle, corpus, golden_frames)
        pc.check_roundtrip(emu_engine, oracle, corpus, big=False)
    finally:
        emu_engine.set_parameter(_lib.PX_SCRATCH_MB, 0)


def test_emu_level9_deep_finder_matches_model(emu_lib_path, oracle, corpus):
    """Level >= 9 launches the deep match finder (two-way far tables on both hashes, live recent-offset rounds, second lazy step): frames
    bit-identical to the model -- on the corpus and on record-like data, whose parse is chains of short repeat-offset matches."""
    import realdata
    from zarc_amd import Engine, _lib
    e9 = Engine(0, lib_path=emu_lib_path)
    try:
        e9.set_parameter(_lib.P_CHECKSUM_FLAG, 1)
        e9.set_parameter(_lib.P_COMPRESSION_LEVEL, 9)
        raws = [corpus.entry(9100 + i, n, i & 3) for i, n in enumerate((0, 70000, 200000, 300001))]
        raws += [bytes(range(200)) * 700, corpus.entry(27, 60000, 0) + corpus.entry(28, 90000, 1) + corpus.entry(27, 60000, 0)]  # joined pieces, far tables (4 ways)
        raws += [realdata.reloc_like(150000), realdata.loglike(140000)]
        for raw, (frame, dig) in zip(raws, e9.pack(raws)):
            assert frame == oracle.zge_encode(raw, oracle.params(level=9))
            assert dig == oracle.blake3(raw)
    finally:
        e9.close()


def test_emu_sequence_stage_with_and_without_the_long_block_kernel(emu_engine, oracle, corpus, golden_frames, monkeypatch):
    """Blocks with long chains that share no tables go to zarc_zdec_seqs_lds (every lane's own tables in LDS), the rest of what the shared-table
    kernel turned down to zarc_zdec_seqs: libzstd's golden frames (a table set per block) and the engine's own take all three routes.
    ZARC_GPU_SEQ_LONG=0 (diagnostic builds only; the emulator library is one) leaves everything to zarc_zdec_seqs: same results."""
    pc.check_unpack_golden(emu_engine, oracle, corpus, golden_frames)
    monkeypatch.setenv("ZARC_GPU_SEQ_LONG", "0")
    pc.check_unpack_golden(emu_engine, oracle, corpus, golden_frames)
    pc.check_roundtrip(emu_engine, oracle, corpus, big=False)
    monkeypatch.setenv("ZARC_GPU_SEQ_SHARED", "0")   # nothing shared: long blocks to the LDS kernel, short ones to the HBM-table kernel
    monkeypatch.setenv("ZARC_GPU_SEQ_LONG", "1")
    pc.check_unpack_golden(emu_engine, oracle, corpus, golden_frames)
    pc.check_roundtrip(emu_engine, oracle, corpus, big=False)


def test_emu_large_frame_is_searched_segment_by_segment(emu_engine, oracle, corpus, libzstds):
    """A frame above ZARC_SPLIT_MIN (4 MiB) is dealt to the match finder's workgroups one 2 MiB segment at a time (zge_match.hip: units);
    the model restarts the finder's carried state at the same places.  Bit-exact, valid for libzstd, round trip."""
    raw = corpus.entry(4242, (4 << 20) + 2300000, 2) [: (4 << 20) + 2300000]
    raw = raw[:3 << 20] + raw[:2 << 20] + raw[3 << 20:]          # a repeat across segment boundaries
    raw = raw[:(4 << 20) + 2300000]
    small = corpus.entry(4243, 70000, 0)
    packed = emu_engine.pack([small, raw, small])
    assert packed[1][0] == oracle.zge_encode(raw)
    assert packed[0][0] == packed[2][0] == oracle.zge_encode(small)
    for z in libzstds:
        assert z.decompress(packed[1][0], len(raw))[0] == raw
    res = emu_engine.unpack([p[0] for p in packed], [len(small), len(raw), len(small)], [p[1] for p in packed])
    assert [r[2] for r in res] == [0, 0, 0] and res[1][0] == raw


def test_emu_frame_pass_in_pieces(emu_engine, oracle, corpus, libzstd15):
    """Frames of 4 MiB and more are cut into pieces wherever nothing in front of a block is read (engine.hip: pieces): libzstd's own
    frames of text (every block reads the 2 MiB before it: one piece), of incompressible data (raw blocks: a piece every 8 blocks) and
    the engine's (independent 2 MiB segments) must all come back bit-exact, together with small frames in the same batch."""
    text = corpus.entry(5151, (4 << 20) + 700000, 0)
    rnd = corpus.entry(5152, (4 << 20) + 300001, 3)
    small = corpus.entry(5153, 90000, 1)
    frames = [libzstd15.compress(text, 3, 1), libzstd15.compress(rnd, 3, 1), emu_engine.pack([text])[0][0], libzstd15.compress(small, 3, 1)]
    raws = [text, rnd, text, small]
    res = emu_engine.unpack(frames, [len(r) for r in raws], [oracle.blake3(r) for r in raws])
    for raw, (out, dig, st) in zip(raws, res):
        assert st == _lib.FRAME_OK and out == raw
    # a corrupted large frame is reported, the others are not touched
    bad = bytearray(frames[0]); bad[len(bad) // 2] ^= 0x55
    res = emu_engine.unpack([bytes(bad), frames[1]], [len(text), len(rnd)], [oracle.blake3(text), oracle.blake3(rnd)])
    assert res[0][2] != _lib.FRAME_OK and res[1][2] == _lib.FRAME_OK and res[1][0] == rnd
    # one large frame among many small ones (round 4, lean sizing: offsets are prefix sums made on the device, only the large frame gets a
    # piece list, the small frames behind it are made up by the frame kernel) -- alone, and dealt into two size groups
    tiny = [corpus.entry(5200 + i, 40 + 37 * i, i % 4) for i in range(70)]
    tframes = [f for f, _ in emu_engine.pack(tiny)]
    order = list(range(70))
    mixed_f = tframes[:30] + [frames[2]] + tframes[30:] + [frames[1]]
    mixed_r = tiny[:30] + [text] + tiny[30:] + [rnd]
    for g in (0, 2):
        emu_engine.set_parameter(_lib.PX_DEC_GROUPS, g)
        try:
            res = emu_engine.unpack(mixed_f, [len(r) for r in mixed_r], [oracle.blake3(r) for r in mixed_r])
        finally:
            emu_engine.set_parameter(_lib.PX_DEC_GROUPS, 0)
        for raw, (out, dig, st) in zip(mixed_r, res):
            assert st == _lib.FRAME_OK and out == raw and dig == oracle.blake3(raw), (g, len(raw))


def test_emu_zero_copy_for_pinned_caller_memory(emu_lib_path, oracle, corpus, golden_frames, monkeypatch):
    """Host-pointer entry points with page-locked caller buffers (engine.hip: segs_pinned / direct_copy): the staging ring is skipped,
    the DMA copies run straight between the caller's memory and the device arenas.  The emulator reports every address as pinned
    when HIPEMU_ALL_PINNED=1; results must equal the staged path's, in one chunk and in many, and ZARC_GPU_PX_ZERO_COPY=0 switches
    the path off."""
    from zarc_amd import Engine
    monkeypatch.setenv("HIPEMU_ALL_PINNED", "1")
    e = Engine(0, emu_lib_path)
    e.set_parameter(_lib.P_CHECKSUM_FLAG, 1)
    e.set_parameter(_lib.PX_ZERO_COPY, 1)      # runs of 1 KiB and more go direct (default: 4 MiB, more than the emulator's test entries)
    try:
        pc.check_roundtrip(e, oracle, corpus, big=False)
        e.set_parameter(_lib.PX_STAGE_CHUNK, 30000)
        pc.check_roundtrip(e, oracle, corpus, big=False)
        pc.check_unpack_errors(e, oracle, corpus, golden_frames)
        ents = [corpus.entry(60 + i, 20000 + 3000 * i, -1) for i in range(6)] + [b""]
        direct = e.pack(ents)
        e.set_parameter(_lib.PX_ZERO_COPY, 0)
        assert e.pack(ents) == direct
    finally:
        e.close()


def test_emu_hash_first_dedup(emu_engine, oracle, corpus):
    """zarc_gpu_pack_batch_dedup (content_frame.rs:26-33: hash, look the digest up, compress only new content): the callback is asked
    once per entry in index order; known content and later copies inside the batch come back as FRAME_DUPLICATE with no frame; the
    frames of what IS compressed equal the plain pack's; across calls the caller's set carries on.  Also through many staged chunks."""
    a, b, c = corpus.entry(80, 30000, 0), corpus.entry(81, 70000, 1), corpus.entry(82, 5000, 2)
    ents = [a, b, a, c, b, b"", a, b""]
    plain = emu_engine.pack(ents)
    for chunk in (0, 20000):
        emu_engine.set_parameter(_lib.PX_STAGE_CHUNK, chunk)
        try:
            seen = set()
            res = emu_engine.pack_dedup(ents, seen)
            assert [r[2] for r in res] == [0, 0, _lib.FRAME_DUPLICATE, 0, _lib.FRAME_DUPLICATE, 0, _lib.FRAME_DUPLICATE, _lib.FRAME_DUPLICATE]
            for (frame, dig, st), (pf, pd), raw in zip(res, plain, ents):
                assert dig == pd == oracle.blake3(raw)
                assert frame == (pf if st == 0 else None)
            assert seen == {oracle.blake3(x) for x in (a, b, c, b"")}
            again = emu_engine.pack_dedup([c, corpus.entry(83, 100, 0)], seen)          # the set carries over to the next call
            assert again[0][2] == _lib.FRAME_DUPLICATE and again[1][2] == 0 and again[1][0] == oracle.zge_encode(corpus.entry(83, 100, 0))
            assert emu_engine.pack_dedup([a, a], {oracle.blake3(a)}) == [(None, oracle.blake3(a), _lib.FRAME_DUPLICATE)] * 2   # nothing to compress at all
        finally:
            emu_engine.set_parameter(_lib.PX_STAGE_CHUNK, 0)


def test_emu_many_tiny_entries(emu_engine, oracle, corpus):
    """More than 4096 entries in one call: the O(n) size ordering of large batches (engine.hip: order_by_size_desc), slot sizes that
    follow the sub-batch's largest block, frames too small for the far table.  Frames equal the model's, round trip, caller's order."""
    import random
    rnd = random.Random(4)
    ents = [corpus.entry(3000 + i, rnd.choice((0, 1, 7, 20, 33, 64, 200)), i % 3) for i in range(4300)] + [corpus.entry(9, 70000, 0), corpus.entry(10, 3000, 1)]
    res = emu_engine.pack(ents)
    for i in list(range(0, 4302, 97)) + [4300, 4301]:
        assert res[i][0] == oracle.zge_encode(ents[i]) and res[i][1] == oracle.blake3(ents[i]), i
    out = emu_engine.unpack([f for f, _ in res], [len(e) for e in ents], [d for _, d in res])
    assert all(st == 0 and o == e for e, (o, d, st) in zip(ents, out))


def test_emu_many_frames_with_turned_down_ones(emu_engine, oracle, corpus):
    pc.check_many_frames_with_turned_down_ones(emu_engine, oracle, corpus, 1500)   # 2 emulated CUs: 64 slots per trip from 512 frames on
