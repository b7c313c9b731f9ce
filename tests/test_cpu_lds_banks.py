"""The LDS layouts of the bf16-plane kernels (matcha_amd/csrc/bf16x3.hpp; fused_bwd.hip, tail_bwd.hip: embed_dim 64; enc128.hip: embed_dim 128)
against the gfx950 bank rules (MI355X guide, section LDS): a wave64 access is served in fixed lane groups, one LDS cycle per group when no
two lanes of a group touch the same bank at different addresses.

  ds_read_b128        bank = (byte / 4) mod 64, 4 banks per lane, groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32 for the upper half)
  ds_read_b64_tr_b16  bank = (byte / 4) mod 64, 2 banks per lane, groups = the two 32-lane halves

Row fragments (contraction over features): lane (c16, kq) reads 16 bytes at row c16, bf16 column 8 kq of a plane.
Column fragments (contraction over tokens): lane l of a 16-lane group kq supplies the address of row base(kq) + ((l & 15) >> 2), bf16 column
4 (l & 3); the second read of the fragment sits HI rows below.  Round 5 measured what the first layout cost (SQ_LDS_BANK_CONFLICT 70 M of 221 M
LDS cycles per launch of the embed_dim-64 backward); this test pins the reasoning behind the layouts that replaced it."""
import itertools

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]
TR_GROUPS = [list(range(32)), list(range(32, 64))]


def worst_way(groups, addr_of, width_banks):
    """Largest number of DISTINCT addresses that meet on one bank inside one lane group (1 = conflict-free)."""
    worst = 1
    for g in groups:
        per_bank = {}
        for lane in g:
            a = addr_of(lane)
            for b in range(width_banks):
                per_bank.setdefault((a // 4 + b) % 64, set()).add(a)
        worst = max(worst, max(len(v) for v in per_bank.values()))
    return worst


def row_read(ps_bf16):
    return lambda lane: ((lane & 15) * ps_bf16 + 8 * (lane >> 4)) * 2


def col_read(ps_bf16, rows_of_group, second=0):
    def addr(lane):
        kq, l = lane >> 4, lane & 15
        return ((rows_of_group(kq) + second + (l >> 2)) * ps_bf16 + 4 * (l & 3)) * 2
    return addr


def test_embed_dim_64_planes_are_conflict_free():
    ps = 80                                                   # fused_bwd.hip / tail_bwd.hip: 40-dword plane rows
    assert worst_way(B128_GROUPS, row_read(ps), 4) == 1
    for second in (0, 16):                                    # slot 8 kq + j <-> token 4 kq + j (j < 4) or 16 + 4 kq + (j - 4)
        assert worst_way(TR_GROUPS, col_read(ps, lambda kq: 4 * kq, second), 2) == 1


def test_embed_dim_128_planes_are_conflict_free():
    ps = 144                                                  # enc128.hip: 72-dword plane rows
    assert worst_way(B128_GROUPS, row_read(ps), 4) == 1
    for second in (0, 16):
        assert worst_way(TR_GROUPS, col_read(ps, lambda kq: 4 * kq, second), 2) == 1


def test_the_first_layouts_were_not():
    # 36-dword rows with token slots 8 kq + j (embed_dim 64, the round's first layout): both reads 2-way
    assert worst_way(B128_GROUPS, row_read(72), 4) == 2
    assert worst_way(TR_GROUPS, col_read(72, lambda kq: 8 * kq), 2) == 2
    # 68-dword rows (embed_dim 128, first layout): row reads 2-way on one slot, transposed reads 2-way
    assert worst_way(B128_GROUPS, row_read(136), 4) == 2
    assert worst_way(TR_GROUPS, col_read(136, lambda kq: 8 * kq), 2) == 2
    # the permuted slots alone do not help on 36-dword rows, the stride alone does not help the transposed reads
    assert worst_way(TR_GROUPS, col_read(72, lambda kq: 4 * kq), 2) >= 2
    assert worst_way(TR_GROUPS, col_read(80, lambda kq: 8 * kq), 2) >= 2
