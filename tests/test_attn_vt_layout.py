"""The index arithmetic of the matrix-core split attention over the transposed value cache (calm_amd/csrc/kernels.hip.h: attn_vt_offset,
k_attn_vt; prefill.hip.h: k_pf_attn_mfma<.., VT>), restated in Python and checked for the properties the kernels rely on -- no GPU
involved.  A slip in one of these formulas otherwise only shows as wrong logits on the device.

* attn_vt_offset: [kv head][block of 64 / ebytes positions][dim][position in block] is a bijection onto the cache, a tile's V^T rows are
  one contiguous 8 KiB, a lane's 16-byte piece is 8 (binary16) / 16 (e5m2) consecutive positions of one head dimension.
* k_attn_vt's key order: A row n of row block rb of sub-tile j is key 8 NT (n >> 2) + 8 j + 4 rb + (n & 3), so that the 8 scores a
  lane holds (C rows 4 kb + e of both row blocks) are the 8 consecutive positions 8 NT kb + 8 j .. + 7 -- what its V^T piece holds.
* its K-image swizzle is injective over the 16 keys one operand fetch touches (per bank phase for 128-byte e5m2 rows).
* k_pf_attn_mfma<VT>: A row r of the 32 x 32 S^T product holds key (r with bits 2 and 3 exchanged); the 8 scores lane-half hh
  contributes to k-step u are keys 16 u + 8 hh .. + 7.
* the order of the summation over the head dimension for an e5m2 cache (k-step t, k-block kb, element e <-> dim) covers every dim once.
"""
import itertools

import pytest

VT_BLOCK_BYTES = 64
HD = 128


def attn_vt_offset(row, pos, head_dim, seq_len, ebytes):
    pb = VT_BLOCK_BYTES // ebytes
    kvh, d = divmod(row, head_dim)
    return ((kvh * (seq_len // pb) + pos // pb) * head_dim + d) * pb + pos % pb


@pytest.mark.parametrize("ebytes", [2, 1])
@pytest.mark.parametrize("n_kv,seq_len", [(1, 64), (2, 192), (3, 448)])
def test_transposed_cache_layout_is_a_bijection_with_contiguous_tiles(ebytes, n_kv, seq_len):
    seen = {}
    for row in range(n_kv * HD):
        for pos in range(seq_len):
            o = attn_vt_offset(row, pos, HD, seq_len, ebytes)
            assert o not in seen
            seen[o] = (row, pos)
    assert sorted(seen) == list(range(n_kv * HD * seq_len))  # exactly the size of the [position][dim] cache: no holes
    pb = VT_BLOCK_BYTES // ebytes  # positions per block = keys per tile of k_attn_vt
    for kvh, blk in itertools.product(range(n_kv), range(seq_len // pb)):
        base = attn_vt_offset(kvh * HD, blk * pb, HD, seq_len, ebytes)
        assert base * ebytes % 8192 == 0  # a tile's V^T rows: one contiguous, aligned 8 KiB ...
        for d, p in itertools.product(range(HD), range(pb)):
            assert attn_vt_offset(kvh * HD + d, blk * pb + p, HD, seq_len, ebytes) == base + d * pb + p  # ... [dim][position in block]
    # the epilogues write dims d and d + 1 of one position: VT_BLOCK_BYTES / ebytes elements apart
    assert attn_vt_offset(5, 7, HD, seq_len, ebytes) + pb == attn_vt_offset(6, 7, HD, seq_len, ebytes)


@pytest.mark.parametrize("nt", [1, 2])
def test_k_attn_vt_key_order_gives_every_lane_consecutive_positions(nt):
    tile = 32 * nt
    keys = set()
    for j in range(nt):
        # S^T's C layout (16 x 16): column = lane & 15 (the query), rows 4 kb + e of row block rb; A row n holds key_of(j, rb, n)
        key_of = lambda rb, n: 8 * nt * (n >> 2) + 8 * j + 4 * rb + (n & 3)
        for kb in range(4):
            held = [key_of(rb, 4 * kb + e) for rb in (0, 1) for e in range(4)]  # slot 4 rb + e of the lane's P operand
            assert held == list(range(8 * nt * kb + 8 * j, 8 * nt * kb + 8 * j + 8))  # = what its V^T piece holds for sub-tile j
            keys.update(held)
    assert keys == set(range(tile))  # every key of the tile is scored exactly once


@pytest.mark.parametrize("kvb", [16, 8])
def test_k_image_swizzle_is_conflict_free_for_an_operand_fetch(kvb):
    nt = 1 if kvb == 16 else 2
    kswz = (lambda k: ((k >> 3) << 2) | (k & 3)) if kvb == 16 else (lambda k: ((k >> 4) << 1) | ((k >> 1) & 1))
    row_bytes = HD * kvb // 8
    for j, rb in itertools.product(range(nt), (0, 1)):
        keys = [8 * nt * (n >> 2) + 8 * j + 4 * rb + (n & 3) for n in range(16)]
        for chunk in range(row_bytes // 16):
            # byte address of the 16-byte chunk each of the 16 lanes (fixed kb, one LDS pass per 8 lanes) reads
            addr = [k * row_bytes + ((chunk ^ kswz(k)) * 16) for k in keys]
            for half in (addr[:8], addr[8:]):
                banks = [(a // 4) % 64 for a in half]
                assert len(set(banks)) == 8  # eight 16-byte pieces of one pass: eight different bank groups


def test_e5m2_head_dimension_order_covers_every_dim_once():
    # k-step t, k-block kb, element e <-> head dimension: a lane's 16-byte chunk (4 u + kb) of a K row holds its operands of k-steps 2 u, 2 u + 1
    dims = [64 * (t >> 1) + 16 * kb + 8 * (t & 1) + e for t in range(4) for kb in range(4) for e in range(8)]
    assert sorted(dims) == list(range(HD))
    for u, kb in itertools.product(range(2), range(4)):
        chunk = [64 * ((2 * u + h) >> 1) + 16 * kb + 8 * h + e for h in (0, 1) for e in range(8)]
        assert chunk == list(range(16 * (4 * u + kb), 16 * (4 * u + kb) + 16))  # chunk index 4 u + kb, low half first


def test_prompt_attention_key_permutation():
    perm = lambda r: (r & ~12) | ((r & 8) >> 1) | ((r & 4) << 1)  # bits 2 and 3 exchanged
    assert sorted(perm(r) for r in range(32)) == list(range(32))
    for u, hh in itertools.product((0, 1), (0, 1)):
        # S^T's C layout (32 x 32): register i of lane-half hh is row (i & 3) + 8 (i >> 2) + 4 hh; operand u is registers 8 u .. 8 u + 7
        rows = [((8 * u + e) & 3) + 8 * ((8 * u + e) >> 2) + 4 * hh for e in range(8)]
        assert [perm(r) for r in rows] == list(range(16 * u + 8 * hh, 16 * u + 8 * hh + 8))
        # ... and the kernel's closed form of the same key
        for e in range(8):
            i = 8 * u + e
            assert perm(rows[e]) == (i & 3) + 4 * ((i >> 2) & 1) + 16 * (i >> 3) + 8 * hh
