"""Host-side check of the index maps of the 16x16x32 conv form (fdsr_conv_k32.hip), lane by lane in numpy -- no GPU:
  * the XOR-swizzled, unpadded halo rows are bank-conflict free for every ds_read_b128 of an activation fragment (the four 16-lane
    groups of the instruction, MI355X_MICROARCH.md section LDS; all kx shifts, both pixel halves, both planes), and the padded-row
    alternative that would not fit the LDS is not;
  * a weight fragment read from the EXISTING arena (pack_weights_h order [cot][kc16][wn][tap][plane][lane] x 8 halves, the 32x32x16
    B-operand map) by the kernel's 16-byte permutation, an activation fragment read from the swizzled halo image, and the
    v_mfma_f32_16x16x32 operand / result maps (A[i = l & 15][k = 8 (l >> 4) + j], B[k][j = l & 15], D[i = 4 (l >> 4) + r][j = l & 15])
    reproduce a direct 3x3 correlation on one workgroup tile, output channel and pixel in the places the epilogue stores them."""
import numpy as np

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def slot_f16x3(slot, hx):
    return slot ^ (hx & 7)


def slot_bf16(slot, hx):
    return slot ^ ((hx >> 1) & 3)


def worst_conflict(addr):
    """addr(col, g, plane) -> byte offset of the lane's 16 bytes within the halo row image (row term is a multiple of 256 B or
    handled by the caller); returns the worst number of distinct addresses on one 16-byte bank slot within a 16-lane group."""
    worst = 0
    for plane in (0, 1):
        for c0 in range(0, 19):            # kx (0..2) + 16 * pixel half
            for grp in GROUPS:
                banks = {}
                for l in grp:
                    a = addr(c0 + (l & 15), l >> 4, plane)
                    banks.setdefault((a // 16) % 16, set()).add(a)
                worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def test_swizzled_halo_rows_are_conflict_free():
    assert worst_conflict(lambda col, g, pl: col * 128 + 16 * slot_f16x3(g + 4 * pl, col)) == 1
    assert worst_conflict(lambda col, g, pl: col * 64 + 16 * slot_bf16(g, col)) == 1
    # what the swizzle replaces: padded 144-byte rows (2-way, and 2 x 18 x 34 x 144 B does not fit 160 KB), plain 128-byte rows
    assert worst_conflict(lambda col, g, pl: col * 144 + 16 * (g + 4 * pl)) == 2
    assert worst_conflict(lambda col, g, pl: col * 128 + 16 * (g + 4 * pl)) >= 4


def test_lo_plane_is_hi_offset_xor_64():
    for hx in range(34):
        for g in range(4):
            assert 16 * slot_f16x3(g + 4, hx) == (16 * slot_f16x3(g, hx)) ^ 64


def pack_arena(w, WN):
    """pack_weights_h (fdsr_engine.cpp) for one plane, values kept as float: arena[cot][kc16][wn][tap][lane][j]."""
    Cout, Cin = w.shape[:2]
    BN = 32 * WN
    ncot, nk = Cout // BN, Cin // 16
    a = np.zeros((ncot, nk, WN, 9, 64, 8), np.float64)
    for cot in range(ncot):
        for kc in range(nk):
            for wn in range(WN):
                for t in range(9):
                    for l in range(64):
                        co = cot * BN + wn * 32 + (l & 31)
                        for j in range(8):
                            a[cot, kc, wn, t, l, j] = w[co, kc * 16 + 8 * (l >> 5) + j, t // 3, t % 3]
    return a


def test_fragment_maps_reproduce_the_convolution():
    rng = np.random.default_rng(0)
    TH, WN = 8, 4                       # one of the kernel's shapes: WM = 2, MB = 4
    WM, BN, MB, HWD = 8 // WN, 32 * WN, TH // (8 // WN), 34
    Cin, Cout = 64, 128
    x = rng.standard_normal((TH + 2, HWD, Cin))          # halo tile (activated input), [hy][hx][c]
    w = rng.standard_normal((Cout, Cin, 3, 3))
    arena = pack_arena(w, WN)
    ref = np.zeros((TH, 32, Cout))
    for ky in range(3):
        for kx in range(3):
            ref += np.einsum('yxc,oc->yxo', x[ky:ky + TH, kx:kx + 32, :], w[:, :, ky, kx])
    out = np.full((TH, 32, Cout), np.nan)
    for wave in range(8):
        wn, wm = wave % WN, wave // WN
        acc = np.zeros((MB, 2, 2, 64, 4))                # [row][pixel half][cout half][lane][reg]
        for kc in range(Cin // 32):
            # halo image of this 32-channel chunk as the staging writes it: byte offset -> 8 halves, f16x3 hi plane only
            lds = {}
            for hy in range(TH + 2):
                for hx in range(HWD):
                    for q in range(8):                   # thread's float4 slot q: channels 4q .. 4q + 3
                        off = (hy * HWD + hx) * 128 + 16 * slot_f16x3(q >> 1, hx) + 8 * (q & 1)
                        lds[off] = x[hy, hx, kc * 32 + 4 * q: kc * 32 + 4 * q + 4]
            for tap in range(9):
                ky, kx = tap // 3, tap % 3
                for mb in range(MB):
                    for ph in range(2):
                        for ch in range(2):
                            A = np.zeros((16, 32))       # weights: rows = couts
                            B = np.zeros((32, 16))       # activations: columns = pixels
                            for l in range(64):
                                g, c15 = l >> 4, l & 15
                                unit = 32 * (g & 1) + 16 * ch + c15              # the kernel's wlane + 16 ch
                                A[c15, 8 * g: 8 * g + 8] = arena[0, 2 * kc + (g >> 1), wn, tap, unit]
                                hx = c15 + kx
                                base = (wm * HWD + hx) * 128 + 16 * slot_f16x3(g, hx)          # xoff[kx][0]
                                off = base + ((mb * WM + ky) * HWD + 16 * ph) * 128           # the ds_read immediate
                                B[8 * g: 8 * g + 8, c15] = np.concatenate([lds[off], lds[off + 8]])
                            D = A @ B
                            for l in range(64):
                                for r in range(4):
                                    acc[mb, ph, ch, l, r] += D[4 * (l >> 4) + r, l & 15]
        for mb in range(MB):
            for ph in range(2):
                for ch in range(2):
                    for l in range(64):
                        g, c15 = l >> 4, l & 15
                        oy, ox = wm + mb * WM, 16 * ph + c15
                        cob = wn * 32 + 4 * g + 16 * ch                         # cot = 0
                        assert np.isnan(out[oy, ox, cob:cob + 4]).all()          # every output written exactly once
                        out[oy, ox, cob:cob + 4] = acc[mb, ph, ch, l]
    assert not np.isnan(out).any()
    np.testing.assert_allclose(out, ref, rtol=1e-10, atol=1e-10)


def test_stats_butterfly_leaves_value_c15_in_lane_c15():
    """The halving butterfly of the epilogue (15 shuffles): lane c holds value index c summed over the 16 pixel lanes."""
    rng = np.random.default_rng(1)
    vals = rng.standard_normal((16, 16))          # [lane][value]
    v = vals.copy()
    half = 8
    while half >= 1:
        nv = v.copy()
        for lane in range(16):
            up = (lane & half) != 0
            for i in range(half):
                keep = v[lane, i + half] if up else v[lane, i]
                send_partner = v[lane ^ half, i] if ((lane ^ half) & half) else v[lane ^ half, i + half]
                nv[lane, i] = keep + send_partner
        v = nv
        half >>= 1
    for lane in range(16):
        np.testing.assert_allclose(v[lane, 0], vals[:, lane].sum(), rtol=1e-12)
