"""CPU model of the fused attention backward's ownership schedule (cmgan_amd/csrc/train.hip, at_bwd_fused_kernel): a
wave owns WRAPPED tile diagonals t and, in round r, works on tile (i, j) = ((r + 2t) mod nbp, (r + t) mod nbp) with
nbp = nb | 1.  The kernel's correctness without atomics rests on four facts that are pure index algebra, checked here for
every block count it accepts (the GPU tests hold the kernel's numbers to the reference's autograd):
  1. every real tile (i, j), i, j < nb, is visited exactly once;
  2. within a round no two slots share a query block or a key block (their dq / dk / dv accumulations cannot collide);
  3. a wave's band-gradient accumulators always belong to ONE real diagonal at a time, and the flush bookkeeping
     (first visit of a real diagonal stores its slab, a later visit adds) writes every one of the 2 nb - 1 slabs;
  4. the launch shape (slots per wave, waves per block) fits the kernel's limits."""
import pytest

ATF_MAX_NB = 22


def atf_step(r, k, nw, wv, nb, nbp):
    """atf_step<SLOTS> of the kernel, verbatim."""
    t = wv + k * nw
    i, j = r + 2 * t, r + t
    i -= nbp if i >= nbp else 0
    i -= nbp if i >= nbp else 0
    j -= nbp if j >= nbp else 0
    valid = t < nbp and i < nb and j < nb
    return (i, j, i - j, t) if valid else None


@pytest.mark.parametrize("nb", range(1, ATF_MAX_NB + 1))
def test_wrapped_diagonal_schedule(nb):
    nbp = nb | 1
    slots = 1 if nbp <= 8 else 2
    nw = (nbp + slots - 1) // slots
    assert nw <= (8 if slots == 1 else 12)                      # __launch_bounds__(512 / 768)
    seen = {}
    slabs = {}                                                  # real diagonal -> list of ("store" | "add")
    state = {(wv, k): dict(cur=None, pos=False, neg=False) for wv in range(nw) for k in range(slots)}

    def flush(st):
        if st["cur"] is None:
            return
        done = st["pos"] if st["cur"] >= 0 else st["neg"]
        slabs.setdefault(st["cur"], []).append("add" if done else "store")
        st["pos" if st["cur"] >= 0 else "neg"] = True

    for r in range(nbp):
        rows, cols = set(), set()
        for wv in range(nw):
            for k in range(slots):
                s = atf_step(r, k, nw, wv, nb, nbp)
                if s is None:
                    continue
                i, j, delta, t = s
                assert (i, j) not in seen, "tile visited twice"
                seen[(i, j)] = (r, wv, k)
                assert i not in rows and j not in cols, "two slots of one round share a query or key block"
                rows.add(i)
                cols.add(j)
                assert (i - j) % nbp == t % nbp
                st = state[(wv, k)]
                if st["cur"] != delta:
                    flush(st)
                    st["cur"] = delta
    for st in state.values():
        flush(st)
    assert set(seen) == {(i, j) for i in range(nb) for j in range(nb)}
    assert set(slabs) == set(range(-(nb - 1), nb))              # all 2 nb - 1 band slabs are written ...
    for delta, ops in slabs.items():
        assert ops[0] == "store" and all(o == "add" for o in ops[1:]), (delta, ops)   # ... first by a store
        assert len(ops) <= 2                                    # pos -> neg -> pos or neg -> pos -> neg
    # a real diagonal is owned by ONE (wave, slot): its stores / adds come from the same wave in program order
    owner = {}
    for (i, j), (r, wv, k) in seen.items():
        assert owner.setdefault(i - j, (wv, k)) == (wv, k)


def test_lds_budget_of_the_largest_launch():
    """three transposed [16][16 nbp + 4] accumulators + two wave-private patches per wave, under the 160 KB of a CU"""
    AT_PB, AT_PS = 36, 48
    for nb in range(1, ATF_MAX_NB + 1):
        nbp = nb | 1
        slots = 1 if nbp <= 8 else 2
        nw = (nbp + slots - 1) // slots
        lds = (3 * 16 * (16 * nbp + 4) + nw * (16 * AT_PB + 16 * AT_PS)) * 4
        assert lds <= 160 * 1024, (nb, lds)
        assert (16 * nbp + 4) % 4 == 0                          # a lane's four rows are one aligned b128
