"""Lane-level numpy emulation of fdsr_conv_wino.hip's index maps (no GPU here): halo staging, the input-transform thread
roles and V image addresses, the A/B fragment maps, accumulator layout, the Z exchange and the final (pixel, cout quad)
pass -- one workgroup, one 16-channel chunk per loop -- against a direct fp64 3x3 correlation.  Also prices the numerics of
the split-f16 Winograd form against the split-f16 direct form on random data."""
import numpy as np

HW, TT, ROWB, RAWB, ZROWB = 18, 64, 80, 80, 272
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)


def pack_u(w):            # w [Cout][Cin][3][3] -> frag[cot][kc][role][nu][lane][8] (plane hi only, fp64)
    Cout, Cin = w.shape[:2]
    U = np.einsum('ak,oikl,bl->oiab', G, w.astype(np.float64), G)       # [co][ci][xi][nu]
    ncw, nk = Cout // 64, Cin // 16
    fr = np.zeros((ncw, nk, 8, 4, 64, 8))
    for cot in range(ncw):
        for kc in range(nk):
            for role in range(8):
                for nu in range(4):
                    for l in range(64):
                        xi, co = role & 3, cot * 64 + (role >> 2) * 32 + (l & 31)
                        for j in range(8):
                            k = kc * 16 + 8 * (l >> 5) + j
                            fr[cot, kc, role, nu, l, j] = U[co, k, xi, nu]
    return fr


def run_wg(x, w, oy0, ox0, cot):
    """x [H][W][Cin] activated input (zero padded outside), returns out[16][16][64] for the workgroup."""
    H, W, Cin = x.shape
    nk = Cin // 16
    fr = pack_u(w)
    acc = np.zeros((8, 4, 2, 64, 16))          # [wave][nu][tb][lane][i]
    for kc in range(nk):
        raw = {}
        for tid in range(512):
            q, row0 = tid & 3, tid >> 2
            for i in range(3):
                pix = row0 + 128 * i
                if pix >= HW * HW:
                    continue
                hy, hx = divmod(pix, HW)
                iy, ix = oy0 - 1 + hy, ox0 - 1 + hx
                v = x[iy, ix, kc * 16 + 4 * q: kc * 16 + 4 * q + 4] if (0 <= iy < H and 0 <= ix < W) else np.zeros(4)
                raw[pix * RAWB + q * 16] = v
        V = {}
        for tid in range(512):
            lane, wave = tid & 63, tid >> 6
            t_tx, t_cq = lane & 7, ((lane >> 3) & 1) | (((lane >> 5) & 1) << 1)
            t_ty, t_xh = (((lane >> 4) & 1) << 2) | (wave & 3), wave >> 2
            t_rd = ((2 * t_ty + t_xh) * HW + 2 * t_tx) * RAWB + t_cq * 16
            t_wr = ((t_xh * 8) * TT + t_ty * 8 + t_tx) * ROWB + t_cq * 8
            R0, R1 = [], []
            for c in range(4):
                a, b, d = (raw[t_rd + (r * HW + c) * RAWB] for r in range(3))
                if t_xh == 0:
                    R0.append(a - d); R1.append(b + d)
                else:
                    R0.append(b - a); R1.append(a - d)
            for r, R in enumerate((R0, R1)):
                outs = (R[0] - R[2], R[1] + R[2], R[2] - R[1], R[1] - R[3])
                for nu, o in enumerate(outs):
                    adr = t_wr + (r * 4 + nu) * (TT * ROWB)
                    assert adr not in V
                    V[adr] = o                  # 4 channels (hi plane; 8 bytes)
        def a_frag(adr):                        # 16-byte read = two 8-byte quads
            return np.concatenate([V[adr], V[adr + 8]])
        for wave in range(8):
            xi, chalf = wave & 3, wave >> 2
            for lane in range(64):
                pass
            for nu in range(4):
                for tb in range(2):
                    A = np.zeros((32, 16)); B = np.zeros((16, 32))
                    for lane in range(64):
                        r31, kh = lane & 31, lane >> 5
                        abase = ((xi * 4) * TT + r31) * ROWB + 16 * kh
                        A[r31, 8 * kh: 8 * kh + 8] = a_frag(abase + (nu * TT + tb * 32) * ROWB)
                        B[8 * kh: 8 * kh + 8, r31] = fr[cot, kc, wave, nu, lane]
                    C = A @ B                   # [tile row m][cout col n]
                    for lane in range(64):
                        r31, kh = lane & 31, lane >> 5
                        for i in range(16):
                            acc[wave, nu, tb, lane, i] += C[(i & 3) + 8 * (i >> 2) + 4 * kh, r31]
    z = {}
    for wave in range(8):
        xi, chalf = wave & 3, wave >> 2
        for lane in range(64):
            r31, kh = lane & 31, lane >> 5
            cz = chalf * 32 + r31
            for tb in range(2):
                for i in range(16):
                    tile = tb * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh
                    m = acc[wave, :, tb, lane, i]
                    z[((xi * 2 + 0) * TT + tile) * ZROWB // 4 + cz] = m[0] + m[1] + m[2]
                    z[((xi * 2 + 1) * TT + tile) * ZROWB // 4 + cz] = m[1] - m[2] - m[3]
    out = np.zeros((16, 16, 64))
    for tid in range(512):
        cqo, pp0 = tid & 15, tid >> 4
        for it in range(8):
            pp = pp0 + it * 32
            py, px = pp >> 4, pp & 15
            tile, i, j = (py >> 1) * 8 + (px >> 1), py & 1, px & 1
            zb = (j * TT + tile) * ZROWB + cqo * 16
            zs = [np.array([z[(zb + (k + i) * 2 * TT * ZROWB) // 4 + e] for e in range(4)]) for k in range(3)]
            out[py, px, cqo * 4: cqo * 4 + 4] = (zs[0] + zs[1]) + zs[2] if i == 0 else (zs[0] - zs[1]) - zs[2]
    return out


def direct(x, w):
    H, W, Cin = x.shape
    xp = np.zeros((H + 2, W + 2, Cin)); xp[1:-1, 1:-1] = x
    out = np.zeros((H, W, w.shape[0]))
    for ky in range(3):
        for kx in range(3):
            out += xp[ky:ky + H, kx:kx + W] @ w[:, :, ky, kx].T.astype(np.float64)
    return out


def split16(v):
    hi = v.astype(np.float16).astype(np.float64)
    lo = (v - hi).astype(np.float16).astype(np.float64)
    return hi, lo


def numerics(seed=0, C=128, n=4096):
    """max error of sum_k a_k w_k per output, direct (9 taps x C) vs Winograd (16 positions), both split-f16 x3 with fp32 accumulate
    emulated in fp64 products rounded per operand only (accumulation error not modelled: fp32 accumulate is common to both)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((18, 18, C)) * 1.5
    w = (rng.random((64, C, 3, 3)) * 2 - 1) / np.sqrt(3 * 9 * C) * np.sqrt(3)
    ref = direct(x, w)[1:-1, 1:-1]
    # direct split
    xh, xl = split16(x); s = 2.0 ** 12
    wh, wl = split16(w * s)
    d3 = (direct(xh, wh) + direct(xh, wl) + direct(xl, wh))[1:-1, 1:-1] / s
    # winograd split: tiles over the interior 16x16
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
    U = np.einsum('ak,oikl,bl->oiab', G, w.astype(np.float64), G)
    su = 2.0 ** np.floor(np.log2(32768.0 / np.abs(U).max()))
    Uh, Ul = split16(U * su)
    outw = np.zeros((16, 16, 64))
    for ty in range(8):
        for tx in range(8):
            d = x[2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4].astype(np.float32)     # patch of the (already padded-by-1) tensor
            Vt = np.einsum('ar,rcx,bc->abx', Bt.astype(np.float32), d, Bt.astype(np.float32)).astype(np.float32)
            Vh, Vl = split16(Vt.astype(np.float64))
            M = (np.einsum('abx,oxab->abo', Vh, Uh) + np.einsum('abx,oxab->abo', Vh, Ul) + np.einsum('abx,oxab->abo', Vl, Uh)) / su
            outw[2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum('ia,abo,jb->ijo', At, M, At)
    scale = np.abs(ref).max()
    return float(np.abs(d3 - ref).max() / scale), float(np.abs(outw - ref).max() / scale)


if __name__ == '__main__':
    rng = np.random.default_rng(1)
    H = W = 32
    Cin, Cout = 32, 128
    x = rng.standard_normal((H, W, Cin))
    w = rng.standard_normal((Cout, Cin, 3, 3)).astype(np.float32)
    ref = direct(x, w)
    worst = 0.0
    for (oy0, ox0, cot) in [(0, 0, 0), (16, 16, 1), (0, 16, 1)]:
        got = run_wg(x, w, oy0, ox0, cot)
        d = np.abs(got - ref[oy0:oy0 + 16, ox0:ox0 + 16, cot * 64:cot * 64 + 64]).max()
        worst = max(worst, d)
        print(f'workgroup ({oy0},{ox0},cot {cot}): max|emulated - direct| = {d:.3e}')
    assert worst < 1e-9
    print('index maps consistent')
    print('split-f16 numerics (direct x3, winograd x3) relative to max|out|:', numerics())
