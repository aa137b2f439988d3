"""The index arithmetic of the prompt-ingestion kernels (calm_amd/csrc/prefill.hip.h), restated in Python and checked for the
properties the kernels rely on -- no GPU involved: these are the formulas in the header's comments, and a slip in one of them is
the kind of bug that otherwise only shows as wrong logits on the device.

* pf_unit: the fragment-major activation matrix ([32-token group][64-column step][MFMA m][hi, lo][lane = (k-half, token)]).
* the key order of the V^T image of k_pf_attn_mfma (P's accumulator registers are the MFMA's B operand as they are).
* the XOR swizzles of the K and V^T images (conflict-free ds_read_b128 for the hardware's 16-lane groups).
* the workgroup order of k_pf_gemm_wide (XCD c takes unit blocks c, c + 8, ...; K ranges; token columns innermost).
"""
import itertools

import numpy as np
import pytest


def pf_unit(t, k, nsteps):
    """16-byte unit of the hi halves of columns k .. k + 7 (k % 8 == 0) of token t; the lo halves are 64 units further"""
    return ((((t >> 5) * nsteps + (k >> 6)) * 4 + ((k & 31) >> 3)) << 7) + (((k >> 5) & 1) << 5) + (t & 31)


@pytest.mark.parametrize("n,tokens", [(64, 32), (4096, 96), (4128, 70), (14336, 64)])
def test_fragment_major_matrix_is_a_bijection_in_lane_order(n, tokens):
    nsteps = (n + 63) // 64
    groups = (tokens + 31) // 32
    seen = {}
    for t in range(groups * 32):
        for k in range(0, nsteps * 64, 8):
            for hl in (0, 1):
                u = pf_unit(t, k, nsteps) + 64 * hl
                assert u not in seen
                seen[u] = (t, k, hl)
    assert sorted(seen) == list(range(groups * nsteps * 512))  # 8 KiB per (group, step): no holes
    # a wave's load for (group, step, MFMA m, hi / lo) is 64 consecutive units: lane = k-half * 32 + token
    for g, s, m, hl in itertools.product(range(groups), (0, nsteps - 1), range(4), (0, 1)):
        base = (((g * nsteps + s) * 4 + m) * 2 + hl) * 64
        for lane in range(64):
            t, k, h = seen[base + lane]
            assert (t, k, h) == (32 * g + (lane & 31), 64 * s + 32 * (lane >> 5) + 8 * m, hl)


def pf_norm_grid(nb, n, ncu):
    """k_pf_norm's grid (prefill.hip.h: pf_norm_grid): 8-token groups x column slices"""
    groups = (nb + 7) // 8
    slices = min(8, (2 * ncu + groups - 1) // groups)
    while slices > 1 and (n >> 3) // slices < 8:
        slices -= 1
    return groups, max(1, slices)


@pytest.mark.parametrize("n,nb", [(4096, 2048), (4096, 128), (2048, 3), (6144, 777), (64, 9), (288, 40)])
def test_norm_kernel_writes_whole_lines_and_covers_every_unit_once(n, nb):
    """k_pf_norm (round 6): a workgroup = 8 consecutive tokens x one column slice, a thread = (token t0 + tid % 8, blocks of 8 columns
    tid / 8, + 32, ...).  What it relies on: the eight tokens' units of the same 8 columns are ONE aligned 128-byte line (so a
    workgroup writes whole lines), and groups x slices x threads cover every (token < nb, column block) exactly once."""
    nsteps = (n + 63) // 64
    for t0, k in itertools.product(range(0, 64, 8), range(0, min(n, 256), 8)):
        units = [pf_unit(t0 + j, k, nsteps) for j in range(8)]
        assert units == list(range(units[0], units[0] + 8)) and units[0] % 8 == 0  # 8 x 16 bytes, line-aligned
    groups, slices = pf_norm_grid(nb, n, 256)
    nkb = n >> 3
    per = (nkb + slices - 1) // slices
    seen = set()
    for g, y, tid in itertools.product(range(groups), range(slices), range(256)):
        t = 8 * g + (tid & 7)
        for kb in range(y * per + (tid >> 3), min(nkb, (y + 1) * per), 32):
            if t < nb:
                assert (t, kb) not in seen
                seen.add((t, kb))
    assert len(seen) == nb * nkb


def mfma_key(u, hh, e):
    """key (within a 32-key tile) that lane-half hh's element e of P operand u stands for: P comes out of S^T's accumulator, whose
    register i of lane-half hh is row (i & 3) + 8 (i >> 2) + 4 hh; operand u is registers 8u .. 8u + 7"""
    i = 8 * u + e
    return (i & 3) + 8 * (i >> 2) + 4 * hh


def vt_position(key):
    """where k_pf_attn_mfma's staging writes `key` in a row of the V^T image"""
    return (key & 16) + 8 * ((key >> 2) & 1) + (key & 3) + 4 * ((key >> 3) & 1)


def test_vt_image_holds_the_keys_in_the_mfma_operand_order():
    assert sorted(vt_position(k) for k in range(32)) == list(range(32))
    for u, hh, e in itertools.product((0, 1), (0, 1), range(8)):
        # the A operand of (u, hh) is the 8 consecutive positions 16 u + 8 hh .. + 7 of the image row
        assert vt_position(mfma_key(u, hh, e)) == 16 * u + 8 * hh + e


# the hardware serves a ds_read_b128 in four groups of 16 lanes (MI355X_MICROARCH.md, LDS table); bank of byte address a: (a / 4) % 64
B128_GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
]


def conflict_free(addr_of_lane):
    for grp in B128_GROUPS:
        banks = set()
        for lane in grp:
            a = addr_of_lane(lane)
            assert a % 16 == 0
            quad = (a // 16) % 16  # a 16-byte access covers four consecutive banks: 16 slots of the 64-bank row
            if quad in banks:
                return False
            banks.add(quad)
    return True


@pytest.mark.parametrize("hd", [64, 128])
def test_k_image_swizzle_is_conflict_free(hd):
    ck = hd // 8  # 16-byte chunks per K row
    kswz = (lambda key: key & 15) if hd == 128 else (lambda key: (key >> 1) & 7)
    for t in range(hd // 16):
        # lane (key = lane & 31, k-half = lane >> 5) reads chunk 2 t + k-half of its key's row
        assert conflict_free(lambda lane: ((lane & 31) * ck + ((2 * t + (lane >> 5)) ^ kswz(lane & 31))) * 16)
    # and the swizzle is a permutation of every row's chunks
    for key in range(32):
        assert sorted(c ^ kswz(key) for c in range(ck)) == list(range(ck))


def test_vt_image_swizzle_is_conflict_free():
    for dt, u in itertools.product(range(4), (0, 1)):
        # lane (d = 32 dt + (lane & 31), hh = lane >> 5) reads the 8-key chunk 2 u + hh of row d (64-byte rows)
        assert conflict_free(lambda lane: ((32 * dt + (lane & 31)) * 4 + ((2 * u + (lane >> 5)) ^ (((32 * dt + (lane & 31)) >> 3) & 3))) * 16)


@pytest.mark.parametrize("rs_bytes,pieces", [(48, 2), (80, 4), (144, 8)])
def test_wide_gemm_a_image_row_stride_is_conflict_free(rs_bytes, pieces):
    """k_pf_gemm_wide's wave-private image of a step of A: rows of 32 / 64 / 128 bytes padded by 16; lane (row = lane & 31,
    k-half = lane >> 5) reads 16-byte piece k-half * P + i of its row (P = pieces / 2)"""
    p = pieces // 2
    for n, i in itertools.product((0, 1), range(p)):
        assert conflict_free(lambda lane: (32 * n + (lane & 31)) * rs_bytes + ((lane >> 5) * p + i) * 16)


@pytest.mark.parametrize("nx,ny,ks", [(16, 16, 1), (112, 4, 1), (24, 16, 1), (16, 4, 4), (5, 3, 2), (125, 16, 1)])
def test_wide_gemm_workgroup_order(nx, ny, ks):
    """one-dimensional grid of 8 * ceil(nx / 8) * ny * ks workgroups -> (unit block, token column, K range): every real tile
    x range exactly once; workgroup id mod 8 (the XCD) == unit block mod 8; within an XCD the token columns of one
    (unit block, range) are consecutive -- the workgroups that share a slice of weights run together on one L2"""
    grid = 8 * ((nx + 7) // 8) * ny * ks
    seen = set()
    per_xcd = {c: [] for c in range(8)}
    for wg in range(grid):
        idx = wg >> 3
        bx, by, k = (wg & 7) + 8 * (idx // (ny * ks)), idx % ny, (idx // ny) % ks
        if bx >= nx:
            continue
        assert bx % 8 == wg % 8
        assert (bx, by, k) not in seen
        seen.add((bx, by, k))
        per_xcd[wg % 8].append((bx, k, by))
    assert len(seen) == nx * ny * ks
    for order in per_xcd.values():
        for a, b in zip(order, order[1:]):
            assert b[:2] == a[:2] and b[2] == a[2] + 1 or b[2] == 0  # next column of the same (block, range), or a new run


def test_k_ranges_cover_every_step_once():
    for nsteps, ks in itertools.product((8, 17, 64, 224), (2, 3, 5, 8)):
        if nsteps < ks:
            continue
        steps = []
        for k in range(ks):
            steps += list(range(k * nsteps // ks, (k + 1) * nsteps // ks))
        assert steps == list(range(nsteps))
