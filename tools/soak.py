#!/usr/bin/env python3
"""Randomised soak of the whole path on a GPU box (not part of the test suite): random sizes / kinds / levels / checksum,
GPU frames against the CPU model (bit-exact), every frame decoded by the oracle, by libzstd and by the GPU decoder, plus
GPU decoding of libzstd's own frames of the same inputs (levels 1..19: Repeat-mode tables, treeless literals, long offsets).
usage: soak.py [seconds] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
from zarc_amd import Engine, _lib
import harness, realdata

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle, corpus, zs = harness.Oracle(), harness.Corpus(), harness.libzstds()
t_end = time.time() + budget
rounds = frames = 0
while time.time() < t_end:
    level = rnd.choice([1, -3, 2, 3, 3, 9, 9, 15])   # all four finder tiers (round 4)
    checksum = rnd.randrange(2)
    eng = Engine(0)
    eng.set_parameter(_lib.P_COMPRESSION_LEVEL, level)
    eng.set_parameter(_lib.P_CHECKSUM_FLAG, checksum)
    eng.set_parameter(_lib.PX_DEC_GROUPS, rnd.choice([0, 0, 1, 2, 3, 4]))   # unpack in size groups (0 = the engine's own rule)
    n = rnd.randrange(1, 40)
    ents = []
    for _ in range(n):
        size = rnd.choice([0, 1, rnd.randrange(2, 400), rnd.randrange(400, 70000), rnd.randrange(70000, 600000), rnd.randrange(600000, 3 << 20),
                           rnd.randrange(3 << 20, 20 << 20) if rnd.randrange(5) == 0 else rnd.randrange(100000, 200000)])
        kind = rnd.randrange(6)
        if kind == 4: raw = realdata.reloc_like(size, seed=rnd.randrange(1 << 30))   # record tables: chains of short repeat-offset matches (the level-9 rounds)
        elif kind == 5: raw = realdata.loglike(min(size, 2 << 20), seed=rnd.randrange(1 << 30))
        else: raw = corpus.entry(rnd.randrange(1 << 30), size, kind)
        if rnd.randrange(6) == 0 and size > 64:      # long runs / repeated halves: RLE blocks, overlapping matches, long matches
            raw = raw[:size // 3] + bytes([raw[0]]) * (size // 3) + raw[:size - 2 * (size // 3)]
        if rnd.randrange(5) == 0 and ents and len(ents[-1]) > 1000:   # far repeats: a mutated copy of the previous entry behind other data (far tables)
            prev = bytearray(ents[-1][:min(len(ents[-1]), 400000)])
            for _ in range(rnd.randrange(0, 30)):
                prev[rnd.randrange(len(prev))] = rnd.randrange(256)
            raw = raw[:len(raw) // 2] + bytes(prev) + ents[-1][:rnd.randrange(0, 100000)]
        if rnd.randrange(8) == 0 and len(raw) > 4096:   # a long exact repeat far back: pieces cut at the compare cap, the continuation guess
            k = rnd.randrange(1000, min(len(raw) // 2, 200000))
            raw = raw + corpus.entry(rnd.randrange(1 << 30), rnd.randrange(70000, 300000), 3) + raw[:k]
        ents.append(raw)
    packed = eng.pack(ents)
    for raw, (frame, dig) in zip(ents, packed):
        assert dig == oracle.blake3(raw)
        if len(raw) <= (10 << 20):                    # bit for bit against the sequential model (frames above 4 MiB: searched by segment)
            p = oracle.params(level=level, checksum=checksum)
            assert frame == oracle.zge_encode(raw, p), (level, checksum, len(raw))
        st, out, used = oracle.zstd_decode(frame, len(raw))
        assert st == 0 and out == raw and used == len(frame)
        for z in zs:
            assert z.decompress(frame, len(raw))[0] == raw
    res = eng.unpack([f for f, _ in packed], [len(e) for e in ents], [d for _, d in packed])
    for raw, (out, dig, st) in zip(ents, res):
        assert st == 0 and out == raw
    # libzstd's own frames through the GPU decoder
    if zs:
        z = rnd.choice(zs)
        lv = rnd.choice([1, 3, 5, 9, 13, 19])
        their = [z.compress(e, lv, rnd.randrange(2)) for e in ents]
        res = eng.unpack(their, [len(e) for e in ents], [oracle.blake3(e) for e in ents])
        for raw, (out, dig, st) in zip(ents, res):
            assert st == 0 and out == raw, (z.version, lv, len(raw))
    eng.close()
    rounds += 1
    frames += n
    if rounds % 50 == 0: print("... %d rounds, %d entries" % (rounds, frames), flush=True)  # the GPU runner kills silent jobs
print("soak ok: %d rounds, %d entries, %.0f s" % (rounds, frames, budget))
