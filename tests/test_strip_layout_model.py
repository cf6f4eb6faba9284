"""CPU model of the index algebra of csrc/conv_strip_kernel.cuh, both strip heights (no GPU, no arithmetic): the packed weight order, the lane -> byte
mapping of the LDS-DMA pieces, the XOR swizzle and the fragment read addresses are restated here with the kernel's own integer
expressions and checked against what the MFMA operands must contain:

  A fragment of lane (l31, lh), row tile mi, tap t, half block hb:  pixel(row 32 mi + l31 shifted by tap t), channels
      32 (hb >> 1) + 16 (hb & 1) + 8 lh + j,  hi at the address, lo one plane further
  B fragment: column 32 ct + l31, the same channels, tap t
and that every ds_read_b128 lane group (MI355X: 4 groups of 16 lanes) touches 16 different 16-byte bank slots.
It is the review-time proof of the layout; the numerical proof is tests/test_gpu_conv.py on the GPU."""
import itertools

import pytest

SHALO, SPW = 4, 16
SHW = SPW + 2
SM, SPH, SHR, SMI = 160, 10, 12 * SHW, 5          # set per test by geom(): 32 SMI rows per strip (csrc/conv_strip_kernel.cuh: SMI = 5 or 1)


def geom(smi):
    global SM, SPH, SHR, SMI
    SMI, SM, SPH = smi, 32 * smi, 2 * smi
    SHR = (SPH + 2) * SHW
# ds_read_b128 lane groups (MI355X_MICROARCH.md, LDS table)
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS = GROUPS + [[l + 32 for l in g] for g in GROUPS]


def swz(spatial, j):
    return ((j // SHW) & 1) if spatial else ((j >> 3) & 1)


def a_slot_image(spatial, NW, row_pixel):
    """LDS bytes of one activation slot after the DMA pieces of all waves: address -> (pixel, group, part, byte in group)."""
    ARV = SHR if spatial else SM + 2 * SHALO
    AR = (ARV + 31) // 32 * 32
    NPP = AR // 32
    NPIECE = 2 * NPP
    PA = (NPIECE + NW - 1) // NW
    img = {}
    for wave in range(NW):
        for i in range(PA):
            q = min(wave + NW * i, NPIECE - 1)
            plane = q // NPP
            for lane in range(64):
                j = (q - plane * NPP) * 32 + (lane >> 1)
                g = (lane & 1) ^ swz(spatial, j)                  # asrc = (g << 5) + (plane << 4)
                pix = row_pixel(j) if j < ARV else -1
                for t in range(16):
                    img[q * 1024 + lane * 16 + t] = (pix, g, plane, t)
    assert len(img) == AR * 64
    return img, AR * 32


@pytest.mark.parametrize("smi", [5, 1])
@pytest.mark.parametrize("NW", [2, 3, 4])
def test_spatial_activation_fragments(NW, smi):
    geom(smi)
    H, W = 23, 37
    py0, px0 = SPH, 16

    def row_pixel(j):
        hy = j // SHW
        y, x = py0 - 1 + hy, px0 - 1 + (j - hy * SHW)
        return y * W + x if (j < SHR and 0 <= y < H and 0 <= x < W) else -1

    img, PLANE = a_slot_image(True, NW, row_pixel)

    def addr(mi, lane, tap):            # the kernel's aad[mi][dy == 1] + compile-time tap offset
        l31, lh = lane & 31, lane >> 5
        r = mi * 32 + l31
        hyc, xc = (r >> 4) + 1, (r & 15) + 1
        a0 = ((hyc - 1) * SHW + xc - 1) * 32 + ((lh ^ ((hyc - 1) & 1)) << 4)
        a1 = (hyc * SHW + xc - 1) * 32 + ((lh ^ (hyc & 1)) << 4)
        dy, dx = tap // 3, tap % 3
        return (a1 if dy == 1 else a0) + (2 * SHW * 32 if dy == 2 else 0) + dx * 32

    for tap, mi, lane in itertools.product(range(9), range(SMI), range(64)):
        l31, lh = lane & 31, lane >> 5
        r = mi * 32 + l31
        ad = addr(mi, lane, tap)
        y, x = py0 + (r >> 4) + tap // 3 - 1, px0 + (r & 15) + tap % 3 - 1
        want = y * W + x if (0 <= y < H and 0 <= x < W) else -1
        for part, a in ((0, ad), (1, ad + PLANE)):
            for j in range(8):
                pix, g, plane, t = img[a + 2 * j]
                assert pix == want and g == lh and plane == part and t == 2 * j
    for tap, mi in itertools.product(range(9), range(SMI)):
        for g in GROUPS:
            slots = {(addr(mi, lane, tap) % 256) // 16 for lane in g}
            assert len(slots) == 16, (tap, mi, g, sorted(slots))      # conflict-free with the line-parity swizzle


@pytest.mark.parametrize("smi", [5, 1])
@pytest.mark.parametrize("vertical", [False, True])
def test_linear_activation_fragments(vertical, smi):
    geom(smi)
    H, W, B = 7, 12, 3
    U, V, su, sv = (W, H, 1, W) if vertical else (H, W, W, 1)
    UV, Mtot = U * V, B * U * V
    ARV = SM + 2 * SHALO
    for m0 in (0, SM, 2 * SM if 2 * SM < Mtot else 0):
        def row_pixel(j):
            m = m0 - SHALO + j
            if not (j < ARV and 0 <= m < Mtot):
                return -1
            q, v = divmod(m, V)
            b, u = divmod(q, U)
            return b * UV + u * su + v * sv

        img, PLANE = a_slot_image(False, 4, row_pixel)
        for tap, mi, lane in itertools.product(range(5), range(SMI), range(64)):
            l31, lh = lane & 31, lane >> 5
            r = mi * 32 + l31
            m = m0 + r
            dv = tap - 2
            fv = m % V if m < Mtot else -1000
            ok = 0 <= fv + dv < V
            row = r + SHALO + dv
            ad = row * 32 + ((lh ^ ((row >> 3) & 1)) << 4) if ok else ARV * 32 + (lh << 4)
            if ok:
                q, v = divmod(m, V)
                b, u = divmod(q, U)
                want = b * UV + u * su + (v + dv) * sv      # the neighbour along the fast axis, same line, same image
            else:
                want = -1                                     # the zero row (a padding row of the slot: refilled from the zero page)
            for part, a in ((0, ad), (1, ad + PLANE)):
                pix, g, plane, t = img[a]
                assert pix == want and plane == part and t == 0 and (g == lh or not ok)
    for tap, mi in itertools.product(range(5), range(SMI)):
        for g in GROUPS:
            slots = set()
            for lane in g:
                row = mi * 32 + (lane & 31) + SHALO + tap - 2
                slots.add(((row * 32 + (((lane >> 5) ^ ((row >> 3) & 1)) << 4)) % 256) // 16)
            assert len(slots) == 16


def test_packed_weight_record():
    """pack_strip_kernel's decode, and the B fragment read of lane (l31, lh) from the record as it sits in LDS."""
    TT, ncb, Npad = 9, 3, 128
    nt32 = Npad // 32
    total = ncb * 2 * TT * Npad * 32

    def decode(i):
        j8, pos, n32, part = i & 7, (i >> 3) & 1, (i >> 4) & 31, (i >> 9) & 1
        ctile, st = (i >> 10) % nt32, (i >> 10) // nt32
        tap, hbk = st % TT, st // TT
        g = pos ^ ((n32 >> 3) & 1)
        return (ctile * 32 + n32, (hbk >> 1) * 32 + (hbk & 1) * 16 + g * 8 + j8, tap, part)

    seen = set()
    for step in (0, 5, TT * 2 * ncb - 1):
        for ct, lane in itertools.product(range(nt32), range(64)):
            l31, lh = lane & 31, lane >> 5
            boff = l31 * 32 + ((lh ^ ((l31 >> 3) & 1)) << 4)
            for part, a in ((0, boff), (1, boff + 1024)):
                for j in range(8):
                    i = ((step * nt32 + ct) * 2048 + a) // 2 + j
                    assert i < total
                    hb, tap = step // TT, step % TT
                    assert decode(i) == (ct * 32 + l31, (hb >> 1) * 32 + (hb & 1) * 16 + lh * 8 + j, tap, part)
                    seen.add(i)
    assert len(seen) == 3 * nt32 * 1024           # every half of the three records is read exactly once
    for g in GROUPS:
        slots = {((l & 31) * 32 + (((l >> 5) ^ (((l & 31) >> 3) & 1)) << 4)) % 256 // 16 for l in g}
        assert len(slots) == 16


# ---- the weight ring and the hand-counted s_waitcnt vmcnt(N) of the main loop (r05: the 3-slot ring of the two-wave 3x3 workgroups) ----
def _simulate_ring(TT, NBST, ring3, PAW, NI, NHB, mode0, swap=None):
    """A wave's vector-memory queue (in-order completion, as vmcnt counts it) through prologue and main loop of
    csrc/conv_strip_kernel.cuh, with the kernel's own wait expressions.  Checks, at every step s = (half block hb, tap T):
      * weight record s+1 has landed before its fragments are read (during the MFMAs of step s),
      * activation half block hb+1 has landed before it is published: the barrier of the last tap (DMA path) or the register ->
        LDS store at tap NBST-2 (register path),
      * the slot a record is requested into is the slot of a record whose fragments were consumed a step ago."""
    swap = ring3 if swap is None else swap
    q = []                                             # outstanding requests, oldest first: ("A", hb) x PAW or ("B", s) x 2 NI
    landed = set()

    def issue(kind, idx, n):
        q.extend([(kind, idx)] * n)

    def wait(n):
        while len(q) > n:
            landed.add(q.pop(0))
    S = NHB * TT
    slot_of = {}
    issue("A", 0, PAW), issue("A", 1, PAW)
    for i in range(NBST - 1):
        issue("B", i, 2 * NI)
        slot_of[i] = i
    wait(0)
    b_next = NBST - 1                                  # next record to request (past the end: the last one again -> same counts)
    for hb in range(NHB):
        for T in range(TT):
            s = hb * TT + T
            last = T == TT - 1
            n = (PAW if T == 0 else 0) if ring3 else 2 * NI * (NBST - 3) + (PAW if T <= NBST - 4 else 0)
            wait(n)
            if s + 1 < S:
                assert ("B", s + 1) in landed, f"step {s}: record {s + 1} still in flight behind wait({n})"
            if not mode0 and T == NBST - 2 and hb + 1 < NHB:
                assert ("A", hb + 1) in landed, f"step {s}: half block {hb + 1} stored before it landed"
            sl = ((hb & 1) * TT + T) % NBST

            def issue_b():
                nonlocal b_next
                dst = (sl + NBST - 1) % NBST
                if b_next < S:
                    # the slot being overwritten held record b_next - NBST: read during step b_next - NBST - 1, consumed in the step after
                    assert b_next - NBST <= s - 1, (s, b_next)
                    assert slot_of.get(b_next - NBST, dst) == dst
                    slot_of[b_next] = dst
                issue("B", min(b_next, S - 1) if b_next < S else ("pad", s), 2 * NI)
                b_next += 1
            if last:
                if mode0 and hb + 1 < NHB:
                    assert ("A", hb + 1) in landed, f"step {s}: barrier publishes half block {hb + 1} before it landed"
                if swap:
                    issue_b()
                issue("A", hb + 2 if hb + 2 < NHB else ("pad", s), PAW)
            if not (swap and last):
                issue_b()
            assert len(q) <= 63, "vmcnt is a 6-bit counter"
    return True


@pytest.mark.parametrize("TT,NBST,ring3,PAW,NI", [(9, 6, False, 4, 1), (9, 6, False, 7, 1), (9, 6, False, 7, 2), (5, 5, False, 3, 1), (5, 5, False, 6, 2),
                                                  (9, 3, True, 7, 1), (9, 3, True, 9, 1), (4, 4, False, 4, 1), (4, 4, False, 5, 1), (4, 4, False, 7, 1)])
@pytest.mark.parametrize("mode0", [True, False])
def test_weight_ring_wait_counts(TT, NBST, ring3, PAW, NI, mode0):
    """The counted waits of RS_STEP for every ring the kernel instantiates: 6 slots (3x3), 5 (1x5 / 5x1) and -- r05 -- 4 slots for the
    2x2-tap form of the stride-2 layers and 3 slots for the two-wave 3x3 workgroups, whose last tap requests the weight record BEFORE the activation half block so that the next step can wait
    for the record (vmcnt(PAW)) and leave the half block in flight."""
    assert (2 * TT) % NBST == 0 and NBST - 2 < TT - 1
    assert _simulate_ring(TT, NBST, ring3, PAW, NI, NHB=8, mode0=mode0)


def test_three_slot_ring_needs_the_swapped_request_order():
    """The model is sensitive to what it checks: the 3-slot wait rule (vmcnt(PAW) at the first tap of a half block) with r04's request
    order -- the activation half block BEFORE the weight record at the last tap -- reads a record that is still in flight."""
    with pytest.raises(AssertionError):
        _simulate_ring(9, 3, True, 7, 1, NHB=4, mode0=False, swap=False)


# ---- r05: the stride-2 form (TT = 4): packed order of pack_strip_s2_kernel + the kernel's tap / plane algebra == a stride-2 3x3 convolution ----
def _pack_s2(w, Npad):
    """csrc/conv_strip.hip pack_strip_s2_kernel, index decode restated: -> {(half block, tap, column, k in 0..15): weight} with
    half blocks running over [plane p = 2 py + px][32-channel block][kk]."""
    import numpy as np
    Cout, Cin, kh, _ = w.shape
    ncb = Cin // 32
    planes = 1 if kh == 1 else 4
    nt32 = Npad // 32
    total = planes * ncb * 2 * 4 * Npad * 32
    out = {}
    for i in range(0, total, 1):
        j8, pos, n32, part = i & 7, (i >> 3) & 1, (i >> 4) & 31, (i >> 9) & 1
        if part:                       # (lo plane: the same value's remainder)
            continue
        ctile = (i >> 10) % nt32
        st = (i >> 10) // nt32
        tap, hbk = st % 4, st // 4
        blk4, kk = hbk >> 1, hbk & 1
        plane, blk = blk4 // ncb, blk4 % ncb
        py, px, ty, tx = plane >> 1, plane & 1, tap >> 1, tap & 1
        ky = (1 if ty == 1 else -1) if py == 0 else (0 if ty == 0 else 2)
        kx = (1 if tx == 1 else -1) if px == 0 else (0 if tx == 0 else 2)
        g = pos ^ ((n32 >> 3) & 1)
        ci = blk * 32 + kk * 16 + g * 8 + j8
        n = ctile * 32 + n32
        v = 0.0
        if kh == 1:
            if n < Cout and tap == 3:
                v = w[n, ci, 0, 0]
        elif n < Cout and ky >= 0 and kx >= 0:
            v = w[n, ci, ky, kx]
        out[(hbk, tap, n, g * 8 + j8)] = v
    return out, planes * ncb * 2


@pytest.mark.parametrize("k", [3, 1])
def test_stride2_form_is_the_stride2_convolution(k):
    """What the TT = 4 instantiation computes, restated with its own integer expressions: for every half block hb (16 channels of
    parity plane p = (hb / 2) / ncb) and tap t = 2 ty + tx, the staged tile of plane p is read at offset (ty - 1, tx - 1) of the
    half-resolution grid (RS_READ1: 2x2 taps (dy, dx) in {0, 1}^2 of the 3x3 halo geometry), plane (py, px) pixel (y, x) being
    source pixel (2 y + py, 2 x + px) (RS_ROW_PIXEL with su = 2 W, sv = 2 and the segment's start pixel) -- summed with the packed
    weights it must be torch's stride-2 convolution with padding k // 2."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(0)
    B, Cin, Cout, H, W = 1, 32, 5, 8, 12
    x = rng.standard_normal((B, Cin, H, W))
    w = rng.standard_normal((Cout, Cin, k, k))
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), stride=2, padding=k // 2).numpy()
    pk, nhb = _pack_s2(w, Npad=32)
    ncb = Cin // 32
    U, V = H // 2, W // 2
    out = np.zeros((B, Cout, U, V))
    for hb in range(nhb):
        plane, blk, kk = (hb >> 1) // ncb, (hb >> 1) % ncb, hb & 1
        py, px = plane >> 1, plane & 1
        for t in range(4):
            tn9 = (t >> 1) * 3 + (t & 1)                    # the kernel's tap index in the 3x3 halo geometry
            dy, dx = tn9 // 3 - 1, tn9 % 3 - 1              # offset on the half-resolution grid: {-1, 0}
            for y in range(U):
                for xx in range(V):
                    ys, xs = y + dy, xx + dx
                    if not (0 <= ys < U and 0 <= xs < V):    # (halo rows outside the plane stage zeros)
                        continue
                    src = x[0, blk * 32 + kk * 16: blk * 32 + kk * 16 + 16, 2 * ys + py, 2 * xs + px]
                    for n in range(Cout):
                        wv = np.array([pk[(hb, t, n, c)] for c in range(16)])
                        out[0, n, y, xx] += float(src @ wv)
    assert np.abs(out - ref).max() < 1e-10
