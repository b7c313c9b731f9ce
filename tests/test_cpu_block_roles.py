"""The tile assignment of front_fwd_kernel and the group assignment of front_fwd3_kernel (matcha_amd/csrc/front_fused.hip) restated on the CPU.

Round 5 put the blocks that build the encoder's per-step weight forms INSIDE the front end's launch (block roles).  For large
batches the launch keeps one block per slot of the chip: the first `nprep` blocks build weight forms and then walk tiles like the others
(they start late), so the last, partial round of tiles goes to the blocks WITHOUT a role first.  For small batches the role blocks leave after
their role and the launch has one more block per tile.  The kernel derives its tiles from (blockIdx, gridDim, nprep, prep_walks, ntiles) with
the arithmetic below; the test pins that every tile is walked exactly once, by a block that exists, for any of these sizes -- a skipped or a
doubled tile would be a wrong (or a raced) row of X."""
import itertools

BIG = 0x3FFFFFF


def launch_shape(slots, nprep, max_tiles):
    """launch_front_fwd: (grid, prep_walks)."""
    prep_walks = 1 if (nprep > 0 and max_tiles + nprep > slots) else 0
    grid = slots if prep_walks else nprep + min(max_tiles, slots)
    return max(grid, nprep + 1), prep_walks


def tiles_of_block(block, grid, nprep, prep_walks, ntiles):
    """front_fwd_kernel: the tiles block `block` walks, in order (None: the block only has a role)."""
    if block < nprep and not prep_walks:
        return None
    nlate = nprep if prep_walks else 0
    vb, gf = block - (nprep - nlate), grid - (nprep - nlate)
    q, r_last = divmod(ntiles, gf)
    last_slot = vb - nlate if vb >= nlate else vb - nlate + gf
    tile_last = q * gf + last_slot if last_slot < r_last else BIG
    out, it = [], 0
    while True:
        tile = vb + it * gf if it < q else (tile_last if it == q else BIG)
        if tile >= ntiles:
            return out
        out.append(tile)
        it += 1


def check(slots, nprep, max_tiles, ntiles):
    grid, walks = launch_shape(slots, nprep, max_tiles)
    seen = []
    per_block = {}
    for b in range(grid):
        t = tiles_of_block(b, grid, nprep, walks, ntiles)
        if t is not None:
            seen += t
            per_block[b] = len(t)
    assert sorted(seen) == list(range(ntiles)), (slots, nprep, max_tiles, ntiles)
    return grid, walks, per_block


def test_every_tile_is_walked_exactly_once():
    for slots, nprep in itertools.product((12, 768), (0, 5, 72)):
        if nprep >= slots:
            continue
        for max_tiles in (1, 3, slots - nprep - 1, slots - nprep, slots - nprep + 1, slots, 2 * slots + 7, 5 * slots - 1, 5 * slots):
            if max_tiles < 1:
                continue
            # the kernel reads the tile count from the plan (device memory): anything up to the launch-time bound, including nothing
            for ntiles in sorted({0, 1, max_tiles // 2, max(max_tiles - 1, 0), max_tiles}):
                check(slots, nprep, max_tiles, ntiles)


def test_large_batches_keep_one_block_per_slot_and_spare_the_role_blocks():
    # the metric's batch: 65 536 rows x 3.5 tokens -> 3 584 tiles of 64 on 768 slots, 72 role blocks
    grid, walks, per_block = check(768, 72, 5121, 3584)
    assert (grid, walks) == (768, 1)
    assert max(per_block.values()) == 5                       # 4 full rounds + 512 tiles of a fifth ...
    assert all(per_block[b] == 4 for b in range(72))          # ... none of which goes to a block that built weight forms first
    assert sum(1 for b in range(72, 768) if per_block[b] == 5) == 512


def test_small_batches_add_the_role_blocks_to_one_block_per_tile():
    grid, walks, per_block = check(768, 72, 30, 23)           # the reference's 384-row step
    assert (grid, walks) == (72 + 30, 0)
    assert set(per_block) == set(range(72, 102)) and sum(per_block.values()) == 23


# ---- front_fwd3_kernel (round 6): wave-independent, groups of 16 tokens, rounds over the wavefronts that are walking ----------------------------
def launch_shape3(slots4, nprep, tcap):
    """launch_front_fwd, attr_mode 1: (grid, prep_walks) -- prep_walks = 3: the role blocks join the group list from round 2 on."""
    ngroups = -(-tcap // 16)
    rounds = -(-ngroups // (slots4 * 4))
    prep_walks = 3 if (nprep > 0 and rounds >= 4) else 0
    grid = slots4 if prep_walks else nprep + min(-(-ngroups // 4), slots4)
    return max(grid, nprep + 1), prep_walks


def groups_of_wave(block, wave, grid, nprep, prep_walks, ngroups):
    """front_fwd3_kernel: the 16-token groups wavefront `wave` of block `block` walks, in order (None: the block only has a role)."""
    late = block < nprep
    if late and not prep_walks:
        return None
    W, Wn = grid * 4, grid * 4 - nprep * 4
    r0 = prep_walks - 1 if prep_walks > 0 else 0x3FFFFF
    wv = block * 4 + wave if late else (block - nprep) * 4 + wave

    def group(r):
        if r < r0:
            return BIG if late else r * Wn + wv
        return r0 * Wn + (r - r0) * W + (wv if late else nprep * 4 + wv)

    out, r = [], (r0 if late else 0)
    while True:
        g = group(r)
        if g >= ngroups:
            return out
        out.append(g)
        r += 1


def check3(slots4, nprep, tcap, t_real):
    grid, walks = launch_shape3(slots4, nprep, tcap)
    ngroups = -(-t_real // 16)
    seen, per_wave = [], {}
    for b in range(grid):
        for w in range(4):
            g = groups_of_wave(b, w, grid, nprep, walks, ngroups)
            if g is not None:
                seen += g
                per_wave[(b, w)] = len(g)
    assert sorted(seen) == list(range(ngroups)), (slots4, nprep, tcap, t_real)
    return grid, walks, per_wave


def test_front_fwd3_every_group_is_walked_exactly_once():
    for slots4, nprep in itertools.product((16, 1024), (0, 5, 72)):
        if nprep >= slots4:
            continue
        for tcap in (1, 17, 64 * (slots4 - nprep), 64 * slots4 * 3 + 5, 64 * slots4 * 4, 64 * slots4 * 4 + 1, 64 * slots4 * 7 + 33):
            # the kernel reads the token count from the plan (device memory): anything up to the launch-time bound
            for t_real in sorted({1, tcap // 3 + 1, max(tcap - 1, 1), tcap}):
                check3(slots4, nprep, tcap, t_real)


def test_front_fwd3_role_blocks_start_two_rounds_late_at_the_bench_batch():
    # the metric's batch: 65 536 x 5 + 1 token slots -> 20 481 groups bound, ~229 000 real tokens -> 14 313 groups on 1024 x 4 wavefronts
    grid, walks, per_wave = check3(1024, 72, 65536 * 5 + 1, 229000)
    assert (grid, walks) == (1024, 3)
    role = [per_wave[(b, w)] for b in range(72) for w in range(4)]
    rest = [per_wave[(b, w)] for b in range(72, 1024) for w in range(4)]
    assert max(role) <= max(rest) - 2 + 1 and min(rest) >= 3 and max(rest) == 4      # (the role blocks' wavefronts skip the first two rounds)
