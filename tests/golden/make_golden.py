#!/usr/bin/env python3
"""Regenerates the committed golden vectors under tests/golden/ (run in the build container).

  blake3_kat.json   published BLAKE3 known-answer vectors (input i%251 pattern + three strings); the
                    oracle must reproduce them -- they are NOT produced by the oracle.
  xxh64_kat.json    XXH64(seed 0) from python-xxhash for the (131*i+7)&255 pattern.
  zstd_frames/      frames produced by REAL libzstd builds found on this image (the reference's codec,
                    zstd-sys 2.0.9+zstd.1.5.5 in Cargo.lock:2480, at other pins) + manifest.json with the
                    input recipe and the BLAKE3/length of the expected output.  Inputs are regenerated from
                    recipes (corpus kinds / seeded patterns), so only compressed bytes are stored.
"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "support"))
from harness import Corpus, Oracle, libzstds  # noqa: E402

BLAKE3_KAT = {
    "strings": {"": "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262",
                "abc": "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85",
                "hello world": "d74981efa70a0c880b8d8c1985d075dbcbf679b99a5f9914e5aaf96b831a9e24"},
    "pattern_mod251": {
        "1": "2d3adedff11b61f14c886e35afa036736dcd87a74d27b5c1510225d0f592e213",
        "1023": "10108970eeda3eb932baac1428c7a2163b0e924c9a9e25b35bba72b28f70bd11",
        "1024": "42214739f095a406f3fc83deb889744ac00df831c10daa55189b5d121c855af7",
        "1025": "d00278ae47eb27b34faecf67b4fe263f82d5412916c1ffd97c8cb7fb814b8444",
        "2048": "e776b6028c7cd22a4d0ba182a8bf62205d2ef576467e838ed6f2529b85fba24a",
        "2049": "5f4d72f40d7a5f82b15ca2b2e44b1de3c2ef86c426c95c1af0b6879522563030",
        "3072": "b98cb0ff3623be03326b373de6b9095218513e64f1ee2edd2525c7ad1e5cffd2",
        "4096": "015094013f57a5277b59d8475c0501042c0b642e531b0a1c8f58d2163229e969",
        "4097": "9b4052b38f1c5fc8b1f9ff7ac7b27cd242487b3d890d15c96a1c25b8aa0fb995",
        "8192": "aae792484c8efe4f19e2ca7d371d8c467ffb10748d8a5a1ae579948f718a2a63",
        "8193": "bab6c09cb8ce8cf459261398d2e7aef35700bf488116ceb94a36d0f5f1b7bc3b",
        "16384": "f875d6646de28985646f34ee13be9a576fd515f76b5b0a26bb324735041ddde4",
        "31744": "62b6960e1a44bcc1eb1a611a8d6235b6b4b78f32e7abc4fb4c6cdcce94895c47",
        "102400": "bc3e3d41a1146b069abffad3c0d44860cf664390afce4d9661f7902e7943e085"}}


def recipe_bytes(recipe, corpus):
    kind = recipe["kind"]
    n = recipe["n"]
    if kind == "corpus":
        return corpus.entry(recipe["index"], n, recipe["ckind"])
    if kind == "zeros":
        return bytes(n)
    rnd = random.Random(recipe["seed"])
    if kind == "few":
        return bytes(rnd.choice(b"ab") for _ in range(n))
    if kind == "sparse":
        return bytes(rnd.randrange(256) if i % 7 else 0 for i in range(n))
    if kind == "runs":
        out = bytearray()
        while len(out) < n:
            out += bytes([rnd.randrange(256)]) * rnd.randint(1, 300)
        return bytes(out[:n])
    if kind == "random":
        return bytes(rnd.getrandbits(8) for _ in range(n))
    if kind == "shuffle":
        # `distinct` units of corpus content in a seeded pseudo-random order, every copy with a few byte mutations: long inputs that
        # compress to little, whose matches reach back across many 128 KiB blocks -- as far as the encoder's window allows (2 MiB at
        # level 3, 4 MiB at level 9, 8 MiB at level 19 for these sizes), in frames that carry a Window_Descriptor
        units = [corpus.entry(recipe["index"] + k, recipe["unit"], recipe["ckind"]) for k in range(recipe["distinct"])]
        out = bytearray()
        while len(out) < n:
            u = bytearray(units[rnd.randrange(recipe["distinct"])])
            for _ in range(recipe["mut"]):
                u[rnd.randrange(len(u))] = rnd.randrange(256)
            out += u
        return bytes(out[:n])
    raise ValueError(kind)


RECIPES = {
    "empty": {"kind": "zeros", "n": 0}, "one": {"kind": "few", "n": 1, "seed": 1},
    "text255": {"kind": "corpus", "n": 255, "index": 100, "ckind": 0}, "text256": {"kind": "corpus", "n": 256, "index": 101, "ckind": 0},
    "text300": {"kind": "corpus", "n": 300, "index": 102, "ckind": 0}, "rand64k": {"kind": "random", "n": 65536, "seed": 2},
    "text128k1": {"kind": "corpus", "n": 131073, "index": 103, "ckind": 0}, "zeros1m": {"kind": "zeros", "n": 1 << 20},
    "records200k": {"kind": "corpus", "n": 200000, "index": 105, "ckind": 1}, "lz300k": {"kind": "corpus", "n": 300000, "index": 106, "ckind": 2},
    "few50k": {"kind": "few", "n": 50000, "seed": 3}, "sparse100k": {"kind": "sparse", "n": 100000, "seed": 4},
    "runs150k": {"kind": "runs", "n": 150000, "seed": 5},
}
# long inputs: frames with a window descriptor, back-references of megabytes across dozens of blocks
BIG_RECIPES = {
    "shuf2m5": {"kind": "shuffle", "n": (5 << 20) // 2, "unit": 96 * 1024 + 7, "distinct": 5, "seed": 11, "index": 200, "ckind": 0, "mut": 9},
    "shuf4m": {"kind": "shuffle", "n": 4 << 20, "unit": 128 * 1024 + 333, "distinct": 6, "seed": 12, "index": 210, "ckind": 1, "mut": 14},
    "shuf16m": {"kind": "shuffle", "n": 16 << 20, "unit": 160 * 1024 + 1, "distinct": 7, "seed": 13, "index": 220, "ckind": 0, "mut": 20},
}


def main():
    import xxhash
    o, c = Oracle(), Corpus()
    json.dump(BLAKE3_KAT, open(os.path.join(HERE, "blake3_kat.json"), "w"), indent=1, sort_keys=True)
    xx = {}
    for n in [0, 1, 3, 4, 7, 8, 31, 32, 33, 63, 64, 1000, 65536, 1 << 20]:
        d = bytes((131 * i + 7) & 255 for i in range(n))
        xx[str(n)] = "%016x" % xxhash.xxh64(d, seed=0).intdigest()
    json.dump(xx, open(os.path.join(HERE, "xxh64_kat.json"), "w"), indent=1, sort_keys=True)
    manifest = {"recipes": dict(RECIPES, **BIG_RECIPES), "frames": []}
    fdir = os.path.join(HERE, "zstd_frames")
    for f in os.listdir(fdir):
        os.remove(os.path.join(fdir, f))
    for z in libzstds():
        levels = [1, 3, 9, 19] if z.version.startswith("1.5") else ([3, 19] if z.version == "1.4.8" else [])
        for lvl in levels:
            for name, rec in RECIPES.items():
                data = recipe_bytes(rec, c)
                for ck in ((0, 1) if name in ("text300", "empty") else (1,)):
                    frame = z.compress(data, lvl, ck)
                    fn = "%s_v%s_l%d_c%d.zst" % (name, z.version, lvl, ck)
                    open(os.path.join(fdir, fn), "wb").write(frame)
                    manifest["frames"].append({"file": fn, "recipe": name, "libzstd": z.version, "level": lvl, "checksum": ck,
                                               "raw_len": len(data), "raw_sha256": hashlib.sha256(data).hexdigest(),
                                               "raw_blake3": o.blake3(data).hex()})
    for z in libzstds():
        if not z.version.startswith("1.5"):
            continue
        for name, rec in BIG_RECIPES.items():
            data = recipe_bytes(rec, c)
            for lvl in (3, 9, 19):
                frame = z.compress(data, lvl, 1)
                fn = "%s_v%s_l%d_c1.zst" % (name, z.version, lvl)
                open(os.path.join(fdir, fn), "wb").write(frame)
                manifest["frames"].append({"file": fn, "recipe": name, "libzstd": z.version, "level": lvl, "checksum": 1,
                                           "raw_len": len(data), "raw_sha256": hashlib.sha256(data).hexdigest(),
                                           "raw_blake3": o.blake3(data).hex()})
                print(fn, len(frame), "descriptor %#x" % frame[4], "window byte %#x" % frame[5])
    json.dump(manifest, open(os.path.join(HERE, "zstd_frames", "manifest.json"), "w"), indent=1, sort_keys=True)
    print("frames:", len(manifest["frames"]))


if __name__ == "__main__":
    main()
