"""CPU: the oracle against golden vectors (published BLAKE3 KATs, python-xxhash, real libzstd frames)."""
import hashlib
import json
import os

import pytest

import make_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_blake3_published_vectors(oracle):
    kat = json.load(open(os.path.join(G, "blake3_kat.json")))
    for s, want in kat["strings"].items():
        assert oracle.blake3(s.encode()).hex() == want
    for n, want in kat["pattern_mod251"].items():
        assert oracle.blake3(bytes(i % 251 for i in range(int(n)))).hex() == want, n


def test_blake3_streaming_equals_oneshot(oracle):
    import ctypes
    data = bytes((i * 7 + 3) & 255 for i in range(70001))
    want = oracle.blake3(data)
    lib = oracle.lib
    for step in (1, 63, 64, 65, 1023, 1024, 1025, 4097, 131072):  # frame_iterator.rs:94-103 feeds arbitrary chunks
        h = ctypes.create_string_buffer(4096)
        lib.oracle_blake3_init(h)
        for i in range(0, len(data), step):
            chunk = data[i:i + step]
            lib.oracle_blake3_update(h, chunk, len(chunk))
        out = ctypes.create_string_buffer(32)
        lib.oracle_blake3_finalize(h, out)
        assert out.raw == want, step


def test_xxh64_vectors(oracle):
    kat = json.load(open(os.path.join(G, "xxh64_kat.json")))
    for n, want in kat.items():
        d = bytes((131 * i + 7) & 255 for i in range(int(n)))
        assert "%016x" % oracle.xxh64(d) == want, n
    assert "%016x" % oracle.xxh64(b"") == "ef46db3751d8e999"


def test_zstd_decoder_on_libzstd_golden_frames(oracle, corpus, golden_frames):
    d, m = golden_frames
    cache = {}
    assert len(m["frames"]) >= 60
    for fr in m["frames"]:
        frame = open(os.path.join(d, fr["file"]), "rb").read()
        name = fr["recipe"]
        if name not in cache:
            cache[name] = make_golden.recipe_bytes(m["recipes"][name], corpus)
        raw = cache[name]
        assert hashlib.sha256(raw).hexdigest() == fr["raw_sha256"]
        rc, out, used = oracle.zstd_decode(frame, len(raw))
        assert rc == 0 and used == len(frame) and out == raw, fr["file"]


def test_zstd_decoder_rejects_corruption(oracle, golden_frames):
    d, m = golden_frames
    fr = next(f for f in m["frames"] if f["recipe"] == "records200k" and f["level"] == 3 and f["checksum"] == 1)
    frame = bytearray(open(os.path.join(d, fr["file"]), "rb").read())
    rc, _, _ = oracle.zstd_decode(bytes(frame[:-1]), fr["raw_len"])
    assert rc != 0  # truncated
    frame[-1] ^= 0x55  # checksum trailer
    rc, _, _ = oracle.zstd_decode(bytes(frame), fr["raw_len"])
    assert rc == -5
    frame[0] ^= 1
    rc, _, _ = oracle.zstd_decode(bytes(frame), fr["raw_len"])
    assert rc == -2


def test_live_libzstd_cross_check(oracle, corpus, libzstds):
    if not libzstds:
        pytest.skip("no libzstd on this box")
    for z in libzstds:
        for kind in range(4):
            raw = corpus.entry(900 + kind, 150000 + kind, kind)
            for lvl in (3, 9):
                rc, out, used = oracle.zstd_decode(z.compress(raw, lvl, 1), len(raw))
                assert rc == 0 and out == raw
