"""CPU model of the software-pipelined x3 attention (attn_sp_out_x3_kernel, cmgan_amd/csrc/attn32_x3.hip): one wave's
whole control flow - the stream of (query tile, 64-key chunk) units, front halves running one unit ahead of back halves
across tile seams, every operand register refilled one unit before its use with the index arithmetic of the kernel
(clamped tile / group indices, the clamp-free distance-table addressing as lane offset + scalar offset), the
query-major distance window with its write / read bases and pitch, the -m splat as the initial value of the E q
accumulator, the reference step outside the hot loop - emulated lane by lane in numpy on the 32x32x16 MFMA layout and
checked against dense Shaw attention (src/models/conformer.py:100-133).  The GPU parity tests hold the kernel itself
to the reference goldens; this is the algebra it was written from."""
import numpy as np
import pytest

from test_attn32_tile_model import A, HH, build_images, f16split, lane, mfma32

P = 100           # ASP_P
HI, LO = 12.0, -4.0


class Wave:
    """One wave = one head.  `seqs` = [(qimg, kimg, vimg)] of the consecutive sequences the block's stream may touch."""

    def __init__(self, seqs, rel, max_pos, L, clamp):
        self.seqs = seqs
        self.rel_hi, self.rel_lo = f16split(rel)
        self.max_pos, self.L, self.Lt, self.clamp = max_pos, L, (L + 31) // 32, clamp
        self.lpad = 32 * self.Lt
        self.R = np.full(32 * P, np.nan)
        self.Rw = A * P + 4 * HH
        self.Rr = A * (P - 1) + 32 + 4 * HH
        self.log = []

    # ---- operand fetches (same clamps / offsets as asp_load_*) ----
    def load_e(self, i0, n, t):
        if self.clamp:
            row = np.clip(self.max_pos - (i0 - 64 * n + 32 - 32 * t) + A, 0, 2 * self.max_pos)
        else:
            i0 = min(i0, self.lpad)
            voff = self.max_pos - 32 + A - self.lpad                   # rows; the kernel adds the plane offset
            soff = 64 * n + 32 * t - i0 + self.lpad
            assert (voff >= 0).all() and soff >= 0
            row = voff + soff
            assert (row >= 0).all() and (row <= 2 * self.max_pos).all(), "clamp-free E fetch outside the table"
        r = 2 * self.max_pos - row              # plane row r holds distance max_pos - r (reversed-order planes, api.hip)
        eh = np.stack([self.rel_hi[r, 8 * HH + e] for e in range(8)], 1)
        el = np.stack([self.rel_lo[r, 8 * HH + e] for e in range(8)], 1)
        return eh, el

    def load_k(self, sq, n, jt):
        kt = min(2 * n + jt, self.Lt - 1)
        return self.seqs[sq][1][kt, 0], self.seqs[sq][1][kt, 1]

    def load_v(self, sq, n, g4):
        return self.seqs[sq][2][min(4 * n + g4, 2 * self.Lt - 1)]

    def load_q(self, sq, it):
        it = min(it, self.Lt - 1)
        return self.seqs[sq][0][it, 0], self.seqs[sq][0][it, 1]

    def tile_of(self, G):
        """(sequence, tile) of flattened tile G, clamped to the last tile of the last sequence (AspTile)."""
        G = min(G, len(self.seqs) * self.Lt - 1)
        return G // self.Lt, G % self.Lt

    # ---- pieces ----
    def eq(self, ah, al, acc):
        r = mfma32(ah, self.qh, acc)
        r = mfma32(ah, self.ql, r)
        return mfma32(al, self.qh, r)

    def wwrite(self, t, r):
        for k in range(4):
            for e in range(4):
                self.R[self.Rw + 32 * t + 8 * k + e] = r[:, 4 * k + e]

    def wread(self, nkt):
        sn = []
        for jt in range(nkt):
            x = np.stack([self.R[self.Rr + 32 * jt + 8 * (v >> 2) + (v & 3)] for v in range(16)], 1)
            assert not np.isnan(x).any(), "window read outside the rows written"
            sn.append(x)
        return sn

    def front(self, nkt, negm, tn, nn, lastq, eshare=False):
        """E q - m, window, K q; then the E / K (/ Q) registers are refilled for chunk nn of tile tn = (sequence, tile).
        eshare: that unit is the next chunk of the same tile, whose window tile 0 is this unit's tile 2 - copied."""
        sq, it = tn
        self.R[:] = np.nan
        for t in range(nkt + 1):
            self.wwrite(t, self.eq(self.eh[t], self.el[t], np.repeat(negm[:, None], 16, 1)))
        if eshare:
            want = self.load_e(32 * it, nn, 0)
            assert np.array_equal(want[0], self.eh[2]) and np.array_equal(want[1], self.el[2])
            self.eh[0], self.el[0] = self.eh[2], self.el[2]
        for t in range(1 if eshare else 0, 3):
            self.eh[t], self.el[t] = self.load_e(32 * it, nn, t)
        sn = self.wread(nkt)
        sn = [self.eq(self.kh[jt], self.kl[jt], sn[jt]) for jt in range(nkt)]
        if lastq:
            self.qh, self.ql = self.load_q(sq, it)
        for jt in range(2):
            self.kh[jt], self.kl[jt] = self.load_k(sq, nn, jt)
        return sn

    def back(self, nkt, s, tv, vn):
        psum = np.zeros(64)
        for g in range(2 * nkt):
            p = np.exp2(s[g >> 1][:, 8 * (g & 1):8 * (g & 1) + 8])
            psum += p.sum(1)
            ph, pl = f16split(p)
            self.o = mfma32(self.va[g], ph, self.o)
            self.o = mfma32(self.va[g], pl, self.o)
        self.l += psum
        if vn is not None:
            for g4 in range(4):
                self.va[g4] = self.load_v(tv[0], vn, g4)

    def reference(self, nkt, full, s, j0):
        mx = np.full(64, -np.inf)
        for jt in range(nkt):
            for v in range(16):
                if not full:
                    key = j0 + 32 * jt + 8 * (v >> 2) + 4 * HH + (v & 3)
                    s[jt][:, v] = np.where(key < self.L, s[jt][:, v], -np.inf)
                mx = np.maximum(mx, s[jt][:, v])
        mx = np.maximum(mx, mx[lane ^ 32])
        run = np.maximum(self.run, mx)
        if ((run > HI) | (run < LO)).any():
            self.log.append("reref")
            alpha = np.where(self.l > 0, np.exp2(-run), 1.0)
            for jt in range(nkt):
                s[jt] = s[jt] - run[:, None]
            self.l = self.l * alpha
            self.o = self.o * alpha[:, None]
            self.m = self.m + run
            self.run = np.zeros(64)
        else:
            self.run = run

    def new_tile(self):
        self.m, self.run, self.l = np.zeros(64), np.full(64, -np.inf), np.zeros(64)
        self.o = np.zeros((64, 16))

    def run_block(self, G0, ntl, gs=1):
        """attn_sp_out_x3_kernel for one wave over the flattened tiles G0 + t * gs, t < ntl; returns {G: stash dict}."""
        L = self.L
        nfull, tail = L >> 6, L & 63
        nch = nfull + (1 if tail else 0)
        nktl = 1 if (tail and tail <= 32) else 2
        fulll = tail == 0
        t0 = self.tile_of(G0)
        self.qh, self.ql = self.load_q(*t0)
        self.eh, self.el, self.kh, self.kl, self.va = [None] * 3, [None] * 3, [None] * 2, [None] * 2, [None] * 4
        for t in range(3):
            self.eh[t], self.el[t] = self.load_e(32 * t0[1], 0, t)
        for jt in range(2):
            self.kh[jt], self.kl[jt] = self.load_k(t0[0], 0, jt)
        for g4 in range(4):
            self.va[g4] = self.load_v(t0[0], 0, g4)
        self.new_tile()
        if nch == 1:
            s = self.front(nktl, -self.m, self.tile_of(G0 + gs), 0, True)
        else:
            s = self.front(2, -self.m, t0, 1, False, eshare=True)
        out = {}
        for tl in range(ntl):
            G = G0 + tl * gs
            tc, t1 = self.tile_of(G), self.tile_of(G + gs)
            if nch > 1:
                self.reference(2, True, s, 0)
            else:
                self.reference(nktl, fulll, s, 0)
            ch = 0
            while ch < nch - 2:                                   # hot loop (+ reference outside on drift)
                sn = self.front(2, -self.m, tc, ch + 2, False, eshare=True)
                self.back(2, s, tc, ch + 1)
                s = sn
                ch += 1
                self.reference(2, True, s, 0)
            if ch < nch - 1:
                sn = self.front(nktl, -self.m, t1, 0, True)
                self.back(2, s, tc, nch - 1)
                self.reference(nktl, fulll, sn, 64 * nfull)
                s = sn
            if tl + 1 < ntl:
                zero = np.zeros(64)
                if nch == 1:
                    sn = self.front(nktl, zero, self.tile_of(G + 2 * gs), 0, True)
                else:
                    sn = self.front(2, zero, t1, 1, False, eshare=True)
                self.back(nktl, s, t1, 0)
            else:
                sn = None
                self.back(nktl, s, None, None)
            inv = 1.0 / (self.l + self.l[lane ^ 32])
            oa = np.stack([(self.o[:, r] + self.o[:, 8 + r]) * inv for r in range(4)], 1)
            ob = np.stack([(self.o[:, 4 + r] + self.o[:, 12 + r]) * inv for r in range(4)], 1)
            stash = {}
            for l in range(64):
                i, cq = A[l] >> 4, A[l] & 15
                stash[(i, HH[l] * 16 + cq)] = oa[l]
                stash[(i, (2 + HH[l]) * 16 + cq)] = ob[l]
            out[G] = stash
            self.new_tile()
            s = sn
        return out


def test_window_banks():
    """ASP_P = 100: a ds_write_b128 store group (8 consecutive lanes) covers the 32 banks exactly once; the skewed
    b32 reads of a 32-lane half hit 32 distinct banks."""
    q = np.arange(8)
    banks = ((q * P)[:, None] + np.arange(4)[None, :]) % 32
    assert len(set(banks.ravel())) == 32
    q = np.arange(32)
    assert len(set((q * (P - 1)) % 32)) == 32
    # and the write base is 16-byte aligned for every lane, every quad
    assert ((A * P + 4 * HH) % 4 == 0).all()


def block_tiles(blk, grid, tpb, group, GN):
    """attn_sp_out_x3_kernel's block -> tile map: XCD blk % 8 owns a contiguous range of the stream; inside it a group
    of `group` consecutive blocks takes interleaved tiles (block j: j, j + gs, ...).  Returns (G0, ntl, gs) or None."""
    xcd, li, nbx = blk & 7, blk >> 3, grid >> 3
    grp = li // group
    gs = min(group, nbx - grp * group)
    G0 = (xcd * nbx + grp * group) * tpb + (li - grp * group)
    if G0 >= GN:
        return None
    return G0, min(tpb, (GN - 1 - G0) // gs + 1), gs


@pytest.mark.parametrize("GN,tpb,group", [(35552, 4, 64), (41088, 81, 64), (44, 4, 64), (20, 16, 64), (1000, 3, 5),
                                          (17, 4, 1), (1, 4, 64), (123, 7, 2)])
def test_block_tile_map_is_a_partition(GN, tpb, group):
    """Every tile of the stream belongs to exactly one block, for the bench shapes (35552 tiles x 4, 41088 x 81) and
    ragged ones; group = 1 is the consecutive order."""
    nb = (GN + tpb - 1) // tpb
    grid = (nb + 7) // 8 * 8
    seen = np.zeros(GN, np.int32)
    for blk in range(grid):
        bt = block_tiles(blk, grid, tpb, group, GN)
        if bt is None:
            continue
        G0, ntl, gs = bt
        assert ntl >= 1 and G0 + (ntl - 1) * gs < GN
        seen[G0 + gs * np.arange(ntl)] += 1
        if group == 1:
            assert gs == 1
    assert (seen == 1).all()


@pytest.mark.parametrize("L,max_pos,scale,tpb,nseq,group",
                         [(101, 512, 1.0, 16, 5, 1), (321, 512, 1.0, 4, 2, 1), (65, 512, 1.0, 2, 3, 1),
                          (33, 512, 1.0, 6, 2, 1), (64, 512, 1.0, 3, 4, 1), (128, 512, 1.0, 3, 2, 1),
                          (200, 512, 1.0, 4, 2, 1), (70, 20, 1.0, 6, 2, 1), (96, 40, 6.0, 2, 2, 1),
                          (321, 512, 5.0, 4, 1, 1), (600, 512, 1.0, 4, 1, 1),
                          (101, 512, 1.0, 3, 5, 2), (321, 512, 1.0, 4, 3, 64), (33, 512, 1.0, 3, 9, 2),
                          (64, 512, 1.0, 2, 7, 3), (70, 20, 1.0, 3, 4, 2)])
def test_attn_sp_unit_stream(L, max_pos, scale, tpb, nseq, group):
    """Blocks of tpb tiles of the flattened (sequence, tile) space of nseq sequences, mapped as the kernel maps them
    (group = 1: consecutive tiles; group > 1: the tiles of co-resident blocks interleave, so a block's next tile is
    usually of another sequence: other Q / K / V images, same distance table).  101 / 321: the model's lengths (no hot iteration
    / four of them; tail chunk of two key tiles / of one key; 16 tiles = 4 sequences per block on the frequency axis);
    64 / 128: no tail chunk; 33 / 64: single-chunk tiles (front halves under the previous tile's only back half, the
    unit after it two tiles ahead); 70 / 96 / 600: distances beyond the table (clamped fetches); scale 5 / 6:
    re-reference inside and outside the hot loop."""
    rng = np.random.default_rng(L + tpb)
    Lt = (L + 31) // 32
    i, j = np.arange(L)[:, None], np.arange(L)[None, :]
    rel = rng.standard_normal((2 * max_pos + 1, 16)) * 0.5
    E = rel[np.clip(i - j, -max_pos, max_pos) + max_pos]
    seqs, refs = [], []
    for _ in range(nseq):
        q = rng.standard_normal((L, 16)) * scale
        k = rng.standard_normal((L, 16))
        v = rng.standard_normal((L, 16))
        S = q @ k.T + np.einsum("id,ijd->ij", q, E)
        Pm = np.exp2(S - S.max(1, keepdims=True))
        refs.append((Pm / Pm.sum(1, keepdims=True)) @ v)
        seqs.append(build_images(q, k, v, Lt))
    clamp = L + 96 > max_pos
    GN = nseq * Lt
    logs = []
    grid = ((GN + tpb - 1) // tpb + 7) // 8 * 8
    done = 0
    for blk in range(grid):
        bt = block_tiles(blk, grid, tpb, group, GN)
        if bt is None:
            continue
        w = Wave(seqs, rel, max_pos, L, clamp)
        out = w.run_block(*bt)
        done += len(out)
        logs += w.log
        for G, stash in out.items():
            sq, it = G // Lt, G % Lt
            for blk in range(2):
                for l16 in range(64):
                    c, g = l16 & 15, l16 >> 4
                    tok = 32 * it + 16 * blk + c
                    if tok < L:
                        np.testing.assert_allclose(stash[(blk, l16)], refs[sq][tok, 4 * g:4 * g + 4], rtol=2e-5, atol=2e-5)
    assert done == GN
    if scale > 1:
        assert "reref" in logs
