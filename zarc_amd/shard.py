"""Which device packs (or unpacks) which entry -- the Python statement of zarc::shard_assign
(zarc_amd/host/zarc_host.hpp), used by bench.py and the tests, and the host-side merge that follows the per-device packs.

Frames are independent (a fresh session per frame, crates/zarc/src/encode/content_frame.rs:37-39), so there is no collective:
every device packs its share; the host then walks the entries in ORIGINAL index order, keeps the first frame of every digest
(content_frame.rs:30-33) and gives it the running offset (content_frame.rs:22,45; the archive starts with 12 header bytes,
encode.rs:65,75).  The way back is the same split (a fresh DCtx per frame, decode/zstd_iterator.rs:28-29; the serial loop at
zarc-cli/src/unpack.rs:62-88): the wanted frames are dealt by UNCOMPRESSED bytes, every device decodes its share, `gather` puts the
results back in the caller's order."""


def assign(sizes, g):
    """-> g lists of entry indices (each ascending).  Equal sizes: index mod g.  Mixed sizes: largest first onto the least loaded
    device by bytes (ties: the lower device)."""
    n = len(sizes)
    if g <= 1:
        return [list(range(n))]
    if all(s == sizes[0] for s in sizes):
        return [list(range(d, n, g)) for d in range(g)]
    out, load = [[] for _ in range(g)], [0] * g
    for i in sorted(range(n), key=lambda i: -sizes[i]):   # stable: equal sizes stay in index order
        d = min(range(g), key=lambda d: load[d])
        out[d].append(i)
        load[d] += sizes[i]
    return [sorted(v) for v in out]


def merge(shares, packed, first_offset=12):
    """shares[d] = indices of device d, packed[d] = [(frame_bytes, digest)] in the same order.  -> (archive body bytes, records) where
    records[i] = (offset, length, digest, written) per ORIGINAL index; a later entry with a digest already seen writes nothing."""
    n = sum(len(s) for s in shares)
    by_index = [None] * n
    for idx, res in zip(shares, packed):
        for i, r in zip(idx, res):
            by_index[i] = r
    body, records, seen, offset = [], [], {}, first_offset
    for frame, digest in by_index:
        if digest in seen:
            records.append((seen[digest][0], seen[digest][1], digest, False))
            continue
        seen[digest] = (offset, len(frame))
        records.append((offset, len(frame), digest, True))
        body.append(frame)
        offset += len(frame)
    return b"".join(body), records


def assign_unpack(uncompressed, g):
    """zarc::FrameReader's split: the same sharder over the frames' uncompressed sizes (the decoder's work)."""
    return assign(list(uncompressed), g)


def gather(shares, results):
    """shares[d] = indices of device d, results[d] = that device's per-frame results in the same order -> results by ORIGINAL index."""
    out = [None] * sum(len(s) for s in shares)
    for idx, res in zip(shares, results):
        assert len(idx) == len(res)
        for i, r in zip(idx, res):
            out[i] = r
    return out
