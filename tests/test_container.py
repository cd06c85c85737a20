"""the Zarc container (header, directory, trailer) written and read by the C++ host mirror
(zarc_amd/host/zarc_container.hpp).  The C++ test binary round-trips an archive through the engine; this file then
re-parses the same archive with an independent python reader that follows SPEC.md / the reference's decoder
(crates/zarc/src/decode/open.rs:21-159, decode/directory.rs:55-119) and checks that the whole file is one valid
Zstandard stream (what `zstd --test` does, and SPEC.md's headline property)."""
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- independent mini CBOR reader -----------------
def cbor(b, p=0):
    ib = b[p]; major, ai = ib >> 5, ib & 31; p += 1
    if major == 7:
        assert ai == 22, "only null expected"
        return None, p
    if ai < 24:
        v = ai
    else:
        n = {24: 1, 25: 2, 26: 4, 27: 8}[ai]
        v = int.from_bytes(b[p:p + n], "big"); p += n
    if major == 0:
        return v, p
    if major == 2:
        return bytes(b[p:p + v]), p + v
    if major == 3:
        return b[p:p + v].decode("utf-8"), p + v
    if major == 4:
        out = []
        for _ in range(v):
            x, p = cbor(b, p); out.append(x)
        return out, p
    if major == 5:
        out = {}
        for _ in range(v):
            k, p = cbor(b, p); x, p = cbor(b, p); out[k] = x
        return out, p
    if major == 6:
        x, p = cbor(b, p)
        return ("tag", v, x), p
    raise AssertionError("unexpected major %d" % major)


def parse_archive(img, decode_frame, blake3):
    """open.rs + directory.rs, restated.  decode_frame(frame_bytes, raw_len) -> bytes."""
    assert img[:8] == bytes.fromhex("502A4D1804000000") and img[8:11] == bytes.fromhex("65AADC") and img[11] == 1
    ep = img[-22:]
    assert ep[19:22] == bytes.fromhex("65AADC") and ep[18] == 1
    digest_type = ep[0]
    off, usize = struct.unpack("<qQ", ep[1:17])
    digest = img[-54:-22]
    assert img[-62:-54] == bytes.fromhex("5F2A4D18") + struct.pack("<I", 54)      # skippable frame, nibble F
    x = 0 ^ digest_type
    for c in digest + ep:
        x ^= c
    assert x == 0, "trailer check byte"                                              # trailer.rs:98-108
    assert off < 0
    dir_at = len(img) + off
    dir_frame = img[dir_at:len(img) - 62]
    assert dir_frame[:4] == bytes.fromhex("28B52FFD")
    d = decode_frame(dir_frame, usize)
    assert len(d) == usize and blake3(d) == digest                                   # decode/directory.rs:112-117
    editions, files, frames, kinds = [], [], [], []
    p = 0
    while p < len(d):
        kind, n, pad = d[p], d[p + 1] | d[p + 2] << 8, d[p + 3]
        assert pad == 0
        item, end = cbor(d, p + 4)
        assert end == p + 4 + n
        p = end
        kinds.append(kind)
        {1: editions, 2: files, 3: frames}[kind].append(item)
    return dict(editions=editions, files=files, frames=frames, kinds=kinds, directory=d, dir_at=dir_at)


def check_archive(img, contents, oracle, libzstds):
    def dec(frame, raw_len):
        status, out, _ = oracle.zstd_decode(frame, raw_len)
        assert status == 0
        return out
    a = parse_archive(img, dec, oracle.blake3)
    assert a["kinds"][0] == 1 and len(a["editions"]) == 1                            # the edition element comes first
    ed = a["editions"][0]
    assert ed[0] == 1 and ed[2] == 1 and ed[1] == ("tag", 0, "2023-12-26T05:04:03+00:00")
    # frames: offsets are running sums from 12 in write order; lengths end exactly at the directory frame
    frames = sorted(a["frames"], key=lambda f: f[1])
    pos = 12
    raw = b""
    for f in frames:
        assert set(f) == {0, 1, 2, 3, 4} and f[0] == 1 and f[1] == pos and len(f[2]) == 32
        body = dec(img[pos:pos + f[3]], f[4])
        assert oracle.blake3(body) == f[2]
        raw += body
        pos += f[3]
    assert pos == a["dir_at"] and raw == contents
    # files: sorted by pathname (BTreeMap order), each normal file's frame element precedes its first use
    names = [tuple(f[1]) for f in a["files"]]
    def key(n):
        return [(isinstance(c, bytes), c.encode() if isinstance(c, str) else c) for c in n]
    assert names == sorted(names, key=key)
    assert ("data", b"caf\xe9") in names
    by_digest = {f[2]: f for f in a["frames"]}
    seen = set()
    order = []                                                                       # element stream order
    p, d = 0, a["directory"]
    while p < len(d):
        n = d[p + 1] | d[p + 2] << 8
        item, _ = cbor(d, p + 4)
        order.append((d[p], item))
        p += 4 + n
    for kind, item in order:
        if kind == 3:
            seen.add(item[2])
        if kind == 2 and 2 in item:
            assert item[2] in seen and item[2] in by_digest
            assert 7 not in item and item[3] == 0o100644
        if kind == 2 and 2 not in item:
            assert item[7] == [1] and item[3] == 0o040755                             # directory special, target dropped
        if kind == 2:
            assert item[6] == {2: ("tag", 0, "2023-11-14T22:13:%02d.000500+00:00" % (20 + len(item[1])))}
    assert order[-1][0] == 3 and order[-1][1][4] == 777                              # the unreferenced frame comes last
    # the whole archive is a valid Zstandard stream: content ++ directory, skippable frames ignored
    for z in libzstds:
        out, err = z.decompress(img, len(contents) + len(a["directory"]))
        assert err is None, (z.version, err)
        assert out == contents + a["directory"]
    return a


def run(binary, tmp_path, big):
    arc, con = str(tmp_path / "t.zarc"), str(tmp_path / "t.contents")
    out = subprocess.check_output([binary, arc, con, str(big)], timeout=900)
    assert b"container OK" in out
    return open(arc, "rb").read(), open(con, "rb").read()


def test_container_emulated(emu_lib_path, tmp_path, oracle, libzstds):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "host"])
    img, contents = run(os.path.join(ROOT, "tests", "emu", "_build", "container_test"), tmp_path, 9000)
    check_archive(img, contents, oracle, libzstds)


@pytest.mark.gpu
def test_container_gpu(tmp_path, oracle, libzstds):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "zarc_amd", "csrc"), "host"])
    img, contents = run(os.path.join(ROOT, "zarc_amd", "container_test"), tmp_path, 3000000)
    check_archive(img, contents, oracle, libzstds)
