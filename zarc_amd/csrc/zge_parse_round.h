// zarc_amd/csrc/zge_parse_round.h -- stages S4 - S6 of the match finder (zge_match.hip: backward propagation, lazy rule, parse walk),
// textually included into the tile loop of zge_match_body: once with ZGE_FIRST 1 for every tile (far inserts and the all-literals
// shortcut of a tile without any match ride along) and, in the level >= 9 instantiation, once more with ZGE_FIRST 0 inside the rounds
// of the live recent-offset pass.  An include rather than a lambda or a loop around one copy: the level-3 kernel sits exactly at its
// 128-register budget, and either form moved its spills (44 -> 72 / 108 bytes of scratch, 196 -> 221 ms on configs[1]).
// Uses the locals of the tile loop; writes fo / fw (final match per position), msel / mlit (the path's matches / literals per chunk).
            // ---- S4: backward propagation.  A position whose match extends b bytes backwards offers it to the b
            // positions before it (ds_max of score << 6 | 63-k: best score wins, then the nearest source); every position
            // then adopts the best offer if it beats its own match.  Same result as scanning the 8 following positions.
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const uint32_t idx = ZGE_IDX(u);
                const uint32_t len = mw[u] & 0xFFFF, back = (mw[u] >> 16) & 0xFF;
                if (len && back && !(dbg & 8)) {
                    const bool rep = (mw[u] >> 24) & 1;
                    // the first eight distances (all a near candidate or a guess can have) as straight-line predicated code: the loop's
                    // bookkeeping -- compare, mask, two branches per trip -- was twice the work of the offers themselves, 9 % of the kernel
                    const int32_t base = score_of(P, len, mo[u], rep); // the score is linear in the length: + lit_cost per byte
                    const uint32_t lim = back < idx ? back : idx;
#pragma unroll
                    for (uint32_t k = 1; k <= (uint32_t)F_BACK_CAP; k++) {
                        const int32_t sc = base + LITC * (int32_t)k;
                        if (k <= lim && sc > 0) atomicMax(&L.ex[idx - k], ((uint32_t)sc << 6) | (63u - k));
                    }
                    for (uint32_t k = (uint32_t)F_BACK_CAP + 1; k <= lim; k++) { // far candidates reach further back
                        const int32_t sc = base + LITC * (int32_t)k;
                        if (sc > 0) atomicMax(&L.ex[idx - k], ((uint32_t)sc << 6) | (63u - k));
                    }
                }
            }
#if ZGE_FIRST
                ZGE_PROF(12); // (diagnostics: the backward offers; stage 4 below is then the wait for the requests and for the slowest wave)
                if (NFAR) zd::wait_vmem(); // the next tile's entries are in registers before any wave sends this tile's inserts
                ZGE_PROF(13);
#endif
            zd::lds_barrier(); // own matches (a0) and offers (ex) are complete
#if ZGE_FIRST
            ZGE_PROF(4);
            if (NFAR) {
                // far inserts of this tile: every wave has used its lookups (they fed S3), so none of them can see these.  Every
                // 2^far_step_log-th position, into the way of this tile; atomic max = the highest position wins, whatever the order.
                const uint32_t way = (tile / TILE) & (uint32_t)(FAR_WAYS - 1);
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const uint32_t idx = ZGE_IDX(u);
                    const uint32_t p = tile + idx;
                    if (FAR_CDC) { // the far positions of this tile: bucket | check bits known since S1
                        const uint32_t hfc = L.a1[idx];
                        if (hfc != 0xFFFFFFFFu && !(dbg & (64 | 8192)))
                            zd::atomic_max_l2(far_l + (size_t)(hfc >> TAG_BITS), ((((uint32_t)(tile - segbase) + idx + 1) << TAG_BITS)) | (hfc & TAG_MASK));
                    } else
                    if (idx < tcount && p < far_end && (p & far_smask) <= far_rmask && !(dbg & (64 | 8192))) {
                        const uint32_t code = ((uint32_t)(tile - segbase) + idx + 1) << TAG_BITS;
                        const uint32_t hf = hash_far32(p8[u], zd::load_u32(tbb + (uint32_t)(p + 8 + wofs))) >> far_shift;
                        zd::atomic_max_l2(far_l + ((size_t)(hf >> TAG_BITS) * FAR_WAYS + way), code | (hf & TAG_MASK));
                        if (FAR_SHORT) {
                            const uint32_t hg = hash_short32(p8[u], SHORT_BYTES) >> far_shift;
                            zd::atomic_max_l2(far_s + ((size_t)(hg >> TAG_BITS) * FAR_WAYS + way), code | (hg & TAG_MASK));
                        }
                    }
                }
            }
            if (L.ctrl[K_ANY] == 0) {
                // no match anywhere in the tile (incompressible data): the path is all literals, nothing to parse
                const uint32_t start = (uint32_t)((pos > tile ? pos : tile) - tile);
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const uint32_t idx = ZGE_IDX(u);
                    if (idx >= start && idx < tcount && !(dbg & 32)) lit_out[lp + (idx - start)] = (uint8_t)p8[u];
                }
                lp += tcount - start;
                if (tid == 0) { L.ctrl[K_POS] = (uint32_t)(tend - bs); if (CONT_CAP) L.ctrl[K_CEND] = 0xFFFFFFFFu; }
                cold++;
                if (cold >= 2) skip_left = cold >= 4 ? 7u : (1u << (cold - 1)) - 1;
                if (NFAR) zd::wait_vmem(); // the far inserts above are in L2 before the next tile's lookups (rare path: a searched tile without any match)
                continue;
            }
            cold = 0;
#endif
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const uint32_t idx = ZGE_IDX(u);
                const uint32_t offer = L.ex[idx];
                uint32_t blen_ = mw[u] & 0xFFFF, boff = mo[u];
                bool brep = (mw[u] >> 24) & 1;
                if (offer) {
                    const int32_t own = blen_ ? score_of(P, blen_, boff, brep) : 0;
                    if ((int32_t)(offer >> 6) > own) {
                        const uint32_t k = 63u - (offer & 63u);
                        const uint32_t nm = L.a0[idx + k];
                        boff = match_off(nm);
                        blen_ = match_len(nm) + k;
                        brep = match_rep(nm);
                    }
                }
                fo[u] = boff;
                fw[u] = blen_ | ((brep ? 1u : 0u) << 24);
                L.a1[idx] = match_pack(boff, blen_, brep); // final match (a1 held this thread's short candidate until S3)
            }
            zd::lds_barrier();
            ZGE_PROF(5);
            // ---- S5: take flag (one-byte lazy lookahead inside the tile) and successor ----
            bool take[PER];
            uint32_t nx[PER]; // true successor in tile coordinates (may leave the tile)
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const uint32_t idx = ZGE_IDX(u);
                const uint32_t my_len = fw[u] & 0xFFFF;
                take[u] = idx < tcount && my_len != 0;
                if (take[u] && P.lazy && idx + 1 < tcount) {
                    const uint32_t m2 = L.a1[idx + 1];
                    const uint32_t l2 = match_len(m2);
                    if (l2 && score_of(P, l2, match_off(m2), match_rep(m2)) > score_of(P, my_len, fo[u], (fw[u] >> 24) & 1) + F_LAZY_DELTA) take[u] = false;
                }
                if (LAZY2 && take[u] && idx + 2 < tcount) { // two bytes ahead (libzstd's lazy2)
                    const uint32_t m3 = L.a1[idx + 2];
                    const uint32_t l3 = match_len(m3);
                    if (l3 && score_of(P, l3, match_off(m3), match_rep(m3)) > score_of(P, my_len, fo[u], (fw[u] >> 24) & 1) + LAZY2) take[u] = false;
                }
                nx[u] = take[u] ? idx + my_len : idx + 1;
            }
            // ---- S6a: per chunk, the first position outside the chunk reached from every position; for the first chunk of the
            // wave's pair this is carried on through the second chunk (one more shuffle), so ex[] of a pair's first half holds
            // the exit of the whole pair ----
            {
                uint32_t val[PER];
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const uint32_t cbase = (uint32_t)(wave * PER + u) * 64;
                    const uint32_t cend = cbase + 64 < tcount ? cbase + 64 : tcount;
                    val[u] = nx[u];
#pragma unroll
                    for (int r = 0; r < 6; r++) {
                        const uint32_t v2 = zd::shfl(val[u], (int)((val[u] - cbase) & 63));
                        if (val[u] < cend) val[u] = v2;
                    }
                }
                {
                    const uint32_t obase = (uint32_t)(wave * PER + 1) * 64, oend = obase + 64 < tcount ? obase + 64 : tcount;
                    const uint32_t through = zd::shfl(val[1], (int)((val[0] - obase) & 63));
                    if (val[0] >= obase && val[0] < oend) val[0] = through;
                }
#pragma unroll
                for (int u = 0; u < PER; u++) L.ex[ZGE_IDX(u)] = val[u];
            }
            zd::lds_barrier();
            ZGE_PROF(6);
            // ---- S6b/c + S7: chunk entries by a chain through ex[], path marks per chunk, emission ----
            {
                zd::wave_priority<2>(); // a serial chain (LDS hops, then a scalar walk): latency matters here, not throughput
                uint32_t cur = (uint32_t)((pos > tile ? pos : tile) - tile);
                int c = 0;
#pragma unroll
                for (; c < wave; c++) { // pairs of chunks before mine: hop over them (an entry in either half leaves through ex[])
                    const uint32_t pend = (uint32_t)(c * 128 + 128) < tcount ? (uint32_t)(c * 128 + 128) : tcount;
                    if (cur < pend) cur = L.ex[cur];
                }
                cur = zd::uniform(cur);
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const int mychunk = wave * PER + u; // the walk of the first chunk ends at the entry of the second
                    const uint32_t cbase = (uint32_t)mychunk * 64;
                    const uint32_t cend = cbase + 64 < tcount ? cbase + 64 : tcount;
                    // walk the path inside my chunk: literal nodes step by one, so only the selected matches are
                    // visited (uniform loop on the scalar unit, one v_readlane per match)
                    const uint64_t tk = zd::ballot(take[u]);
                    const uint32_t span = cend > cbase ? cend - cbase : 0u; // positions of this chunk inside the tile
                    // The scalar loop only collects the selected matches (find the next take flag at or after the cursor, jump to its
                    // successor): five or six scalar instructions and a v_readlane per match.  The literals are what is left: a position
                    // at or after the chunk's entry that is no selected match and does not lie inside the nearest selected match below
                    // it -- one ds_bpermute for the whole chunk instead of two 64-bit masks built per match.
                    uint64_t sel = 0, lits = 0;
                    const uint32_t entry = cur; // where the path enters this chunk (>= cend: it does not)
                    uint32_t o_last = 0, o_diff = 0; // REP_PASS: offset of the chunk's last selected match, and the last one different from it (0: none)
                    if (!(dbg & 2)) {
                        while (cur < cend) {
                            const uint64_t ahead = tk & (~0ull << (cur - cbase)); // take flags at or after the cursor
                            if (ahead == 0) { cur = cend; break; }                // only literals up to the end of the chunk
                            const uint32_t q = (uint32_t)zd::ctz64(ahead);
                            sel |= 1ull << q;
                            cur = zd::readlane(nx[u], q);
                            if (REP_PASS) { const uint32_t o = zd::readlane(fo[u], q); if (o != o_last) { o_diff = o_last; o_last = o; } }
                        }
                        const uint64_t below = sel & lt;                          // selected matches below this lane
                        const uint32_t qs = below ? 63u - (uint32_t)__clzll((long long)below) : 0u;
                        const uint32_t over = zd::shfl(nx[u], (int)qs);           // where the nearest one below ends
                        const uint32_t me = cbase + (uint32_t)lane;
                        lits = zd::ballot(me >= entry && (uint32_t)lane < span && !((sel >> lane) & 1) && (!below || over <= me));
                    }
                    msel[u] = sel;
                    mlit[u] = lits;
                    if (lane == 0) L.wcnt[mychunk] = ((uint32_t)__popcll(msel[u]) << 16) | (uint32_t)__popcll(mlit[u]);
                    if (REP_PASS && lane == 0) { L.wrep[2 * mychunk] = o_last; L.wrep[2 * mychunk + 1] = o_diff; }
                }
                // the wave that owns the last chunk knows where the path leaves the tile
                if (wave == WAVES - 1 && lane == 0) L.ctrl[K_POS] = (uint32_t)(tile - bs) + cur;
                zd::wave_priority<0>();
            }
            zd::lds_barrier();
            ZGE_PROF(7);
