"""SURVEY.md section 8 rows f2 + f3: the `zarc pack | unpack | list-files` command line over the engine
(zarc_amd/host/zarc_cli.cpp), driven the way crates/zarc-cli is: a directory tree goes in, a `.zarc` comes out, the
tree comes back with contents, modes and timestamps.  The archive is also re-read by the independent python parser of
test_container.py and decoded as one Zstandard stream by every libzstd on the box."""
import os
import re
import subprocess

import pytest

from test_container import parse_archive

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_tree(base, corpus):
    src = base / "src"
    (src / "sub" / "deep").mkdir(parents=True)
    files = {
        "a.txt": corpus.entry(900, 12000, 0),
        "b.bin": corpus.entry(901, 70000, 3),
        "empty": b"",
        "sub/c.txt": corpus.entry(900, 12000, 0),          # same content as a.txt: one frame
        "sub/deep/d.rec": corpus.entry(902, 300000, 1),
    }
    for name, data in files.items():
        p = src / name
        p.write_bytes(data)
        os.chmod(p, 0o640)
        os.utime(p, ns=(1_600_000_000_123_456_000, 1_650_000_000_654_321_000))
    os.symlink("../a.txt", src / "sub" / "link")
    os.chmod(src / "sub", 0o750)
    for d in (src / "sub" / "deep", src / "sub", src):       # deepest first: touching a directory's entries moves its own mtime
        os.utime(d, ns=(1_610_000_000_000_000_000, 1_620_000_000_111_222_000))
    try:                                                   # extended attribute (metadata/encode.rs:343-372); not every file system takes user.*
        os.setxattr(src / "b.bin", "user.zarc.test", b"caf\xc3\xa9 \xff\x00bytes")
        os.setxattr(src / "a.txt", "user.note", b"plain text")
    except OSError:
        pass
    return files


def run_cli(binary, tmp_path, corpus, oracle, libzstds, store=False, gpus=0, env=None):
    files = make_tree(tmp_path, corpus)
    arc = tmp_path / ("out-store.zarc" if store else "out.zarc")
    cmd = [binary, "pack", "--output", str(arc), "--level", "3", "--zstd", "ChecksumFlag=true"] + (["--store"] if store else []) + ["src"]
    out = subprocess.run(cmd, cwd=tmp_path, capture_output=True, timeout=900, check=True)
    m = re.fullmatch(rb"digest: ([A-Za-z0-9+/]{43}=)\n", out.stdout)
    assert m, out.stdout
    digest = m.group(1).decode()
    img = arc.read_bytes()

    # --- the archive, read independently ---
    def dec(frame, raw_len):
        st, o, _ = oracle.zstd_decode(frame, raw_len)
        assert st == 0
        return o
    a = parse_archive(img, dec, oracle.blake3)
    import base64
    assert base64.b64encode(img[-54:-22]).decode() == digest
    names = ["/".join(c if isinstance(c, str) else c.decode() for c in f[1]) for f in a["files"]]
    assert names == sorted(["src", "src/a.txt", "src/b.bin", "src/empty", "src/sub", "src/sub/c.txt", "src/sub/deep", "src/sub/deep/d.rec", "src/sub/link"],
                           key=lambda n: n.split("/"))
    by_name = dict(zip(names, a["files"]))
    assert by_name["src/a.txt"][2] == by_name["src/sub/c.txt"][2] == oracle.blake3(files["a.txt"])
    assert len(a["frames"]) == 4                                                  # a(=c), b, empty, d
    assert by_name["src/sub"][7] == [1] and by_name["src/sub/link"][7] == [10, "../a.txt"] and 2 not in by_name["src/sub/link"]
    assert by_name["src/a.txt"][3] & 0o7777 == 0o640 and by_name["src/sub"][3] & 0o7777 == 0o750
    assert by_name["src/a.txt"][6][2] == ("tag", 0, "2022-04-15T05:20:00.654321+00:00")   # modified
    assert by_name["src/a.txt"][6][3] == ("tag", 0, "2020-09-13T12:26:40.123456+00:00")   # accessed
    assert by_name["src/a.txt"][4][0] == os.getuid() and by_name["src/a.txt"][5][0] == os.getgid()
    try:                                                                          # xattrs: text when valid UTF-8, bytes otherwise (key 12)
        want = {"user.note": "plain text"} if os.getxattr(tmp_path / "src" / "a.txt", "user.note") else None
    except OSError:
        want = None
    if want:
        assert by_name["src/a.txt"][12] == want
        assert by_name["src/b.bin"][12] == {"user.zarc.test": b"caf\xc3\xa9 \xff\x00bytes"}
    total_raw = sum(f[4] for f in a["frames"])
    for z in libzstds:                                                            # `zstd --test` equivalent
        o, err = z.decompress(img, total_raw + len(a["directory"]))
        assert err is None, (z.version, err)
    if store:
        assert all(f[3] == 14 + f[4] + 3 * max(1, -(-f[4] // 131072)) for f in a["frames"])

    # --- list-files ---
    out = subprocess.run([binary, "list-files", str(arc)], capture_output=True, timeout=600, check=True)
    assert out.stdout.decode().splitlines() == [n + ("/" if n in ("src", "src/sub", "src/sub/deep") else "@" if n.endswith("link") else "") for n in names]
    out = subprocess.run([binary, "list", str(arc), "--only-files", "--filter", r"\.txt$"], capture_output=True, timeout=600, check=True)
    assert out.stdout.decode().splitlines() == ["src/a.txt", "src/sub/c.txt"]

    # --- global flags: -v counts up the level, --log-file writes JSON lines (args.rs:39-65, logs.rs:12-67) ---
    out = subprocess.run([binary, "-vv", "list-files", str(arc)], capture_output=True, timeout=600, check=True)
    assert b" INFO zarc: logging initialised" in out.stderr and len(out.stdout.decode().splitlines()) == len(names)
    logdir = tmp_path / "logs"
    logdir.mkdir(exist_ok=True)
    env_log = dict(os.environ); env_log.pop("RUST_LOG", None)
    out = subprocess.run([binary, "--log-file", str(logdir), "pack", "--output", str(tmp_path / "logged.zarc"), "src"], cwd=tmp_path, capture_output=True,
                         timeout=900, check=True, env=env_log)
    logs = list(logdir.glob("zarc.*.log"))
    assert len(logs) >= 1 and re.fullmatch(r"zarc\.\d{4}-\d\d-\d\dT\d\d-\d\d-\d\dZ\.log", logs[0].name)
    import json
    lines = [json.loads(l) for l in logs[0].read_text().splitlines()]
    assert lines[0]["fields"]["message"] == "logging initialised" and any(l["level"] == "DEBUG" for l in lines)   # --log-file alone means -vvv
    for l in logs:
        l.unlink()

    # --- unpack ---
    dest = tmp_path / ("dest-store" if store else "dest")
    dest.mkdir()
    out = subprocess.run([binary, "unpack", str(arc)], cwd=dest, capture_output=True, timeout=900, check=True)
    assert ("digest: %s\n" % digest).encode() in out.stderr and b"unpacked 5 files" in out.stderr
    for name, data in files.items():
        p = dest / "src" / name
        assert p.read_bytes() == data, name
        st = p.stat()
        assert st.st_mode & 0o7777 == 0o640 and st.st_mtime_ns == 1_650_000_000_654_321_000
    assert (dest / "src" / "sub").stat().st_mode & 0o7777 == 0o750
    for d in ("src", "src/sub", "src/sub/deep"):            # directory metadata goes on last, deepest first: files created inside do not clobber it
        assert (dest / d).stat().st_mtime_ns == 1_620_000_000_111_222_000, d
    assert not os.path.lexists(dest / "src" / "sub" / "link")                     # like the reference: links are listed, not recreated
    # --verify
    ok = subprocess.run([binary, "unpack", str(arc), "--verify", digest, "--filter", "b.bin"], cwd=dest, capture_output=True, timeout=600)
    assert ok.returncode == 0 and b"unpacked 1 files" in ok.stderr and b"digest:" not in ok.stderr
    bad = subprocess.run([binary, "unpack", str(arc), "--verify", "A" * 43 + "="], cwd=dest, capture_output=True, timeout=600)
    assert bad.returncode == 1 and b"integrity failure: zarc file digest is " + digest.encode() in bad.stderr
    # --- several devices: the same archive body and the same tree, frames dealt to N engine handles both ways ---
    if gpus > 1:
        arc2 = tmp_path / "out-g.zarc"
        cmd2 = [binary, "pack", "--output", str(arc2), "--level", "3", "--gpus", str(gpus)] + (["--store"] if store else []) + ["src"]
        subprocess.run(cmd2, cwd=tmp_path, capture_output=True, timeout=900, check=True, env=env)
        img2 = arc2.read_bytes()
        assert img2[:a["dir_at"]] == img[:a["dir_at"]]                          # every content frame, byte for byte, at the same offsets
        dest2 = tmp_path / "dest-g"
        dest2.mkdir()
        out = subprocess.run([binary, "unpack", str(arc2), "--gpus", str(gpus)], cwd=dest2, capture_output=True, timeout=900, check=True, env=env)
        assert b"unpacked 5 files" in out.stderr
        for name, data in files.items():
            assert (dest2 / "src" / name).read_bytes() == data, name
        many = subprocess.run([binary, "unpack", str(arc2), "--gpus", "63"], cwd=dest2, capture_output=True, timeout=600, env=env)
        assert many.returncode == 1 and b"--gpus 63" in many.stderr             # more than the box has: a clean error, not a crash
    # --- levels and hints the engine maps or ignores say so once (the reference forwards them all to libzstd, pack.rs:24-33, 86-217) ---
    if not store:
        w = subprocess.run([binary, "pack", "--output", str(tmp_path / "w.zarc"), "--level", "19", "--zstd", "Strategy=btopt", "--zstd", "EnableLongDistanceMatching=true",
                            "--zstd", "WindowLog=20", "src"], cwd=tmp_path, capture_output=True, timeout=900, env=env)
        assert w.returncode == 0, w.stderr[-500:]
        assert b"warning: --level 19 packs with the engine's level-15 finder" in w.stderr
        assert w.stderr.count(b"is accepted but advisory") == 2                 # Strategy and EnableLongDistanceMatching, not WindowLog
        q = subprocess.run([binary, "pack", "--output", str(tmp_path / "q.zarc"), "--level", "3", "src"], cwd=tmp_path, capture_output=True, timeout=900, env=env)
        assert q.returncode == 0 and b"warning" not in q.stderr
        f = subprocess.run([binary, "pack", "--output", str(tmp_path / "f.zarc"), "--level=-5", "src"], cwd=tmp_path, capture_output=True, timeout=900, env=env)
        assert f.returncode == 0 and b"level-1 finder" in f.stderr, f.stderr[-300:]
        dest3 = tmp_path / "dest-f"
        dest3.mkdir()
        subprocess.run([binary, "unpack", str(tmp_path / "f.zarc")], cwd=dest3, capture_output=True, timeout=900, check=True, env=env)
        for name, data in files.items():
            assert (dest3 / "src" / name).read_bytes() == data, name
    # --- errors inside the pipeline end in "Error: ..." and exit code 1, never in an abort (reader / writer threads are joined) ---
    if os.path.exists("/dev/full"):
        full = subprocess.run([binary, "pack", "--output", "/dev/full", "src"], cwd=tmp_path, capture_output=True, timeout=900, env=env)
        assert full.returncode == 1 and b"Error:" in full.stderr, (full.returncode, full.stderr[-500:])
    # a damaged archive is refused
    broken = bytearray(img); broken[-5] ^= 1
    (tmp_path / "broken.zarc").write_bytes(bytes(broken))
    bad = subprocess.run([binary, "list-files", str(tmp_path / "broken.zarc")], capture_output=True, timeout=600)
    assert bad.returncode == 1 and b"check byte" in bad.stderr


def test_cli_emulated(emu_lib_path, tmp_path, corpus, oracle, libzstds):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "host"])
    binary = os.path.join(ROOT, "tests", "emu", "_build", "zarc")
    env2 = dict(os.environ, HIPEMU_DEVICES="2")             # the emulator then reports two devices: `--gpus 2` opens ordinals 0 and 1
    run_cli(binary, tmp_path, corpus, oracle, libzstds, gpus=2, env=env2)
    (tmp_path / "s").mkdir()
    run_cli(binary, tmp_path / "s", corpus, oracle, libzstds, store=True)


def test_cli_emulated_eight_devices(emu_lib_path, tmp_path, corpus):
    """Eight engine handles on one host (BASELINE configs[4] shape in small: sizes a few KiB .. 700 KiB, duplicates whose
    copies land on different devices): `zarc pack --gpus 8` writes the archive `--gpus 1` writes, byte for byte up to the directory
    (whose timestamp differs), `unpack --gpus 8` restores the tree, and a corrupt frame gives the same error either way."""
    import random
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "host"])
    binary = os.path.join(ROOT, "tests", "emu", "_build", "zarc")
    env8 = dict(os.environ, HIPEMU_DEVICES="8")
    rnd = random.Random(8)
    src = tmp_path / "tree"
    src.mkdir()
    files = {}
    for i in range(26):
        n = int(3000 * 2 ** (rnd.random() * 8))
        files["f%02d.bin" % i] = corpus.entry(900 + i, n, i % 4)
    for i in (3, 11, 19):
        files["dup%02d.bin" % i] = files["f%02d.bin" % i]                    # same content, another name: one frame, wherever the copies are dealt
    for name, data in files.items():
        (src / name).write_bytes(data)
    arcs = {}
    for g in (1, 8):
        arc = tmp_path / ("g%d.zarc" % g)
        subprocess.run([binary, "pack", "--output", str(arc), "--gpus", str(g), "tree"], cwd=tmp_path, capture_output=True, timeout=1800, check=True, env=env8)
        arcs[g] = arc.read_bytes()
    # content frames are written in call order with first-wins dedup: everything in front of the directory frame is identical
    def dir_at(img):
        import struct
        off = struct.unpack("<q", img[-22 + 1:-22 + 9])[0]               # epilogue: digest_type u8 | directory_offset i64 (negative, from the end)
        return len(img) + off
    d1, d8 = dir_at(arcs[1]), dir_at(arcs[8])
    assert d1 == d8 and arcs[1][:d1] == arcs[8][:d8]
    for g in (1, 8):
        dest = tmp_path / ("dest%d" % g)
        dest.mkdir()
        out = subprocess.run([binary, "unpack", str(tmp_path / ("g%d.zarc" % 8)), "--gpus", str(g)], cwd=dest, capture_output=True, timeout=1800, check=True, env=env8)
        assert b"unpacked %d files" % len(files) in out.stderr
        for name, data in files.items():
            assert (dest / "tree" / name).read_bytes() == data, (g, name)
    broken = bytearray(arcs[8])
    broken[12 + 40000] ^= 0x10                                              # inside an early content frame
    (tmp_path / "broken.zarc").write_bytes(bytes(broken))
    res = {}
    for g in (1, 8):
        dest = tmp_path / ("bad%d" % g)
        dest.mkdir()
        r = subprocess.run([binary, "unpack", str(tmp_path / "broken.zarc"), "--gpus", str(g)], cwd=dest, capture_output=True, timeout=1800, env=env8)
        res[g] = (r.returncode, sorted(l for l in r.stderr.splitlines() if b"rror" in l or b"corrupt" in l.lower()))
    assert res[1] == res[8] and res[1][0] != 0, res


@pytest.mark.gpu
def test_cli_gpu(tmp_path, corpus, oracle, libzstds):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "zarc_amd", "csrc"), "host"])
    binary = os.path.join(ROOT, "zarc_amd", "zarc")
    from zarc_amd import _lib
    ndev = _lib.load().zarc_gpu_device_count()
    print("test_cli_gpu: %d device(s) visible%s" % (ndev, "" if ndev >= 2 else " -- `--gpus 2` section SKIPPED (needs two devices)"))
    run_cli(binary, tmp_path, corpus, oracle, libzstds, gpus=2 if ndev >= 2 else 0)
    (tmp_path / "s").mkdir()
    run_cli(binary, tmp_path / "s", corpus, oracle, libzstds, store=True)
