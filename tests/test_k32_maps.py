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


# ---------------------------------------------------------------------------------------------------------------------------
# the column-strip form (fdsr_conv_strip.hip): row slots, row tiles, the rolling three-row accumulation
# ---------------------------------------------------------------------------------------------------------------------------
def strip_swz(RB, px):
    return 2 * ((px >> 1) & 3) if RB in (128, 384) else 2 * (px & 7)


def _worst(groups, addr, width):
    worst = 0
    for grp in groups:
        banks = {}
        for l in grp:
            a = addr(l)
            for b in range(a // 4, (a + width) // 4):
                banks.setdefault(b % 64, set()).add(a)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def test_strip_row_slots_are_conflict_free():
    """Every ds_read_b128 of an activation fragment (lane -> pixel c15 + kx + 16 ph, 16-byte unit plane * OPP + 4 kc + g, XOR strip_swz)
    and every ds_write_b128 of a staged oct, for the pixel strides of the instantiations: 128 B (64 channels bf16), 256 (128 ch bf16,
    64 ch f16x3: hi | lo planes, or a 128-channel rider row), 384 (192 channels), 512."""
    for RB in (128, 256, 384, 512):
        units = RB // 16
        for plane_units in ({0} if RB in (128, 384) else {0, units // 2}):
            for kc in range((units - plane_units) // 4 if plane_units == 0 else units // 8):
                for kx in range(3):
                    for ph in range(4):
                        def rd(l):
                            g, c = l >> 4, l & 15
                            px = 16 * ph + c + kx
                            u = (plane_units + 4 * kc + g) ^ strip_swz(RB, px)
                            assert 0 <= u < units
                            return px * RB + 16 * u
                        assert _worst(GROUPS, rd, 16) == 1, (RB, kc, kx, ph)
        opp = units if RB in (128, 384) else units // 2        # octs per pixel and plane (f16x3 / rider rows: see above)
        for opp_t in {8, 16} & {opp, opp // 2, 8}:
            for base in range(0, 64, 256 // opp_t):
                def wr(l, opp_t=opp_t, base=base):             # thread -> (oct tid % opp_t, pixel 1 + tid // opp_t)
                    tid = l
                    px = 1 + base + tid // opp_t
                    return px * RB + 16 * ((tid % opp_t) ^ strip_swz(RB, px))
                assert _worst([list(range(j, j + 8)) for j in range(0, 64, 8)], wr, 16) == 1


def test_strip_row_tiles_are_conflict_free():
    """The output / residual row tiles [pixel][64 couts]: accumulator-side quads (8 bytes per lane in bf16: ds_write_b64 over four
    16-lane groups, ds_read_b64 over two 32-lane groups; 16 bytes in the fp32-output modes) and row-side 16-byte units, under the
    unit XOR tsw(px) = (px >> 1) & 7 (128-byte rows) / 2 (px & 7) (256-byte rows)."""
    for OSZ in (2, 4):
        PB, UPP = 64 * OSZ, 64 * OSZ // 16
        tsw = (lambda px: (px >> 1) & 7) if OSZ == 2 else (lambda px: 2 * (px & 7))
        for w in range(4):
            for ph in range(4):
                def aq(l):
                    g, c = l >> 4, l & 15
                    px = 16 * ph + c
                    if OSZ == 2:
                        return px * PB + 16 * (((4 * w + g) >> 1) ^ tsw(px)) + 8 * (g & 1)
                    return px * PB + 16 * ((4 * w + g) ^ tsw(px))
                if OSZ == 2:
                    assert _worst([list(range(i, i + 16)) for i in range(0, 64, 16)], aq, 8) == 1
                    assert _worst([list(range(0, 32)), list(range(32, 64))], aq, 8) == 1
                else:
                    assert _worst(GROUPS, aq, 16) == 1
                    assert _worst([list(range(j, j + 8)) for j in range(0, 64, 8)], aq, 16) == 1
            for i in range(64 * PB // 16 // 256):
                def a128(l):
                    tid = 64 * w + l
                    px = tid // UPP + (256 // UPP) * i
                    return px * PB + 16 * ((tid % UPP) ^ tsw(px))
                assert _worst(GROUPS, a128, 16) == 1
                assert _worst([list(range(j, j + 8)) for j in range(0, 64, 8)], a128, 16) == 1
        # both sides address the same bytes: unit u of pixel px
        seen = {}
        for w in range(4):
            for ph in range(4):
                for l in range(64):
                    g, c = l >> 4, l & 15
                    px, co = 16 * ph + c, 16 * w + 4 * g
                    a = px * PB + 16 * (((co * OSZ) // 16) ^ tsw(px)) + (co * OSZ) % 16
                    seen[(px, co)] = a
        for tid in range(256):
            for i in range(64 * PB // 16 // 256):
                px, u = tid // UPP + (256 // UPP) * i, tid % UPP
                a = px * PB + 16 * (u ^ tsw(px))
                for e in range(0, 16, 4 * OSZ):                 # the quads inside this unit
                    co = (16 * u + e) // OSZ
                    assert seen[(px, co)] == a + e


def test_strip_rolling_rows_and_fragments_reproduce_the_convolution():
    """One strip segment in numpy: weight fragments read from the arena by the kernel's permutation (wave w = couts 16 w .. 16 w + 15),
    input rows staged one per step into two swizzled slots, every fragment multiplied with the ky = 2, 1, 0 weights into the rotating
    accumulator sets (A2 / A1 / A0 = (ROT + 2, ROT, ROT + 1) % 3), a row leaving its set one step after it was finished -- equals the
    direct 3x3 correlation with zero padding, every output written exactly once."""
    rng = np.random.default_rng(2)
    H, SW, Cin, Cout, WN = 7, 64, 64, 64, 2
    RB, KCH, NPH = Cin * 2, Cin // 32, SW // 16
    x = rng.standard_normal((H, SW + 2, Cin))            # activated rows incl. the two halo columns
    x[:, 0] = 0
    wgt = rng.standard_normal((Cout, Cin, 3, 3))
    arena = pack_arena(wgt, WN)
    xp = np.zeros((H + 2, SW + 2, Cin)); xp[1:-1] = x
    ref = np.zeros((H, SW, Cout))
    for ky in range(3):
        for kx in range(3):
            ref += np.einsum('yxc,oc->yxo', xp[ky:ky + H, kx:kx + SW], wgt[:, :, ky, kx])
    out = np.full((H, SW, Cout), np.nan)
    for w in range(4):
        co32, cot, wna = w >> 1, (w >> 1) // WN, (w >> 1) % WN
        acc = np.zeros((3, NPH, 64, 4))
        oy0 = 0
        for it in range(H + 2):                          # input rows -1 .. H
            iy, rot = oy0 - 1 + it, it % 3
            A0, A1, A2 = (rot + 1) % 3, rot, (rot + 2) % 3
            if it >= 3:                                  # the row finished by the previous step leaves set A0 before its fresh product
                for ph in range(NPH):
                    for l in range(64):
                        g, c = l >> 4, l & 15
                        assert np.isnan(out[iy - 2, 16 * ph + c, 16 * w + 4 * g]).all()
                        out[iy - 2, 16 * ph + c, 16 * w + 4 * g: 16 * w + 4 * g + 4] = acc[A0, ph, l]
            row = xp[iy + 1] if 0 <= iy + 1 < H + 2 else np.zeros((SW + 2, Cin))
            lds = {}
            for px in range(SW + 2):
                for o in range(Cin // 8):
                    lds[px * RB + 16 * (o ^ strip_swz(RB, px))] = row[px, 8 * o: 8 * o + 8]
            for kx in range(3):
                for kc in range(KCH):
                    for ph in range(NPH):
                        B = np.zeros((32, 16))
                        for l in range(64):
                            g, c = l >> 4, l & 15
                            px = c + kx
                            B[8 * g: 8 * g + 8, c] = lds[px * RB + 16 * ((4 * kc + g) ^ strip_swz(RB, px)) + ph * 16 * RB]
                        for ky, a in ((2, A2), (1, A1), (0, A0)):
                            A = np.zeros((16, 32))
                            for l in range(64):
                                g, c = l >> 4, l & 15
                                A[c, 8 * g: 8 * g + 8] = arena[cot, 2 * kc + (g >> 1), wna, ky * 3 + kx, 32 * (g & 1) + 16 * (w & 1) + c]
                            D = A @ B
                            fresh = ky == 0 and kx == 0 and kc == 0
                            for l in range(64):
                                for r in range(4):
                                    acc[a, ph, l, r] = (0.0 if fresh else acc[a, ph, l, r]) + D[4 * (l >> 4) + r, l & 15]
        last = ((H + 2 - 1) % 3 + 2) % 3                 # the last row sits in the last step's ky = 2 set
        for ph in range(NPH):
            for l in range(64):
                g, c = l >> 4, l & 15
                assert np.isnan(out[H - 1, 16 * ph + c, 16 * w + 4 * g]).all()
                out[H - 1, 16 * ph + c, 16 * w + 4 * g: 16 * w + 4 * g + 4] = acc[last, ph, l]
    assert not np.isnan(out).any()
    np.testing.assert_allclose(out, ref, rtol=1e-10, atol=1e-10)
