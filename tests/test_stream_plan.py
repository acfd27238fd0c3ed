"""cmgan_amd.streaming.StreamCursor - the host arithmetic of the carried-state stream (which frames the device buffers hold,
the offsets each step's captured graph bakes in) - replayed on arrays of frame NUMBERS: every stage must see exactly the
frames the contract names (oracle/stream_oracle.py::stream_forward: encoder frames [e0 - 15, e1), TSCB frames
[k W - Ca, e1), kept frames [k W, k W + n_keep), decoder frames [k W - 15, k W + n_keep)), for the one-graph-per-step body
and for the pipelined stages with their parity double buffers, over window / context / look-ahead settings that exercise
every branch (windows shorter than the 15-frame history, no look-ahead, clips shorter than one window, ragged ends)."""
import numpy as np
import pytest

from cmgan_amd.streaming import HIST_FRAMES, StreamCursor

H = HIST_FRAMES


def _steps(T, W, La):
    """(n_new, last) per step, as enhance_stream feeds them."""
    fed, k, out = 0, 0, []
    while fed < T:
        upto = min((k + 1) * W + La, T)
        out.append((upto - fed, upto == T))
        fed, k = upto, k + 1
    return out


def _expect(k, e0, e1, W, Ca, last):
    lo = k * W
    n_keep = e1 - lo if last else min(W, e1 - lo)
    return dict(enc=np.arange(max(e0 - H, 0), e1), tscb=np.arange(max(lo - Ca, 0), e1), kept=np.arange(lo, lo + n_keep),
                dec=np.arange(max(lo - H, 0), lo + n_keep))


@pytest.mark.parametrize("T,W,Ca,La", [(1601, 400, 40, 40), (1601, 400, 40, 0), (161, 40, 12, 8), (97, 8, 20, 3), (50, 64, 8, 8),
                                       (321, 100, 40, 40), (33, 5, 0, 0), (1000, 37, 50, 11)])
def test_one_graph_per_step_body_sees_the_contract_frames(T, W, Ca, La):
    cur = StreamCursor(W, Ca, La)
    cap = H + La + W + La
    S, E, D = np.full(cap + 64, -1), np.full(Ca + La + W + La + 64, -1), np.full(H + W + La + 64, -1)
    emitted = []
    for n_new, last in _steps(T, W, La):
        p = cur.plan(n_new, last)
        want = _expect(p["k"], p["e0"], p["e1"], W, Ca, last)
        new = np.arange(p["e0"], p["e1"])
        # StreamState._graph_body on frame numbers
        S[p["n_tail"]:p["n_tail"] + n_new] = new
        np.testing.assert_array_equal(S[p["n_tail"] - p["h_enc"]:p["n_tail"] + n_new], want["enc"])
        E[p["n_ctx"]:p["n_ctx"] + n_new] = new                          # (the encoder's history outputs are dropped)
        x = E[:p["n_ctx"] + n_new].copy()
        np.testing.assert_array_equal(x, want["tscb"])
        kept = x[p["keep_lo"]:p["keep_lo"] + p["n_keep"]]
        np.testing.assert_array_equal(kept, want["kept"])
        D[p["h_dec"]:p["h_dec"] + p["n_keep"]] = kept
        np.testing.assert_array_equal(D[:p["h_dec"] + p["n_keep"]], want["dec"])
        np.testing.assert_array_equal(S[p["dec_lo"]:p["dec_lo"] + p["h_dec"] + p["n_keep"]], want["dec"])   # the decoder's spectrogram frames
        emitted.append(kept.copy())
        if not last:
            for buf, lo, hi in ((S, p["spec_drop"], p["n_tail"] + n_new), (E, p["enc_drop"], p["n_ctx"] + n_new),
                                (D, p["h_dec"] + p["n_keep"] - p["h_dec_next"], p["h_dec"] + p["n_keep"])):
                if lo > 0 and hi > lo:
                    buf[:hi - lo] = buf[lo:hi].copy()
        assert p["n_tail"] + n_new <= cap and p["n_ctx"] + n_new <= Ca + La + W + La and p["h_dec"] + p["n_keep"] <= H + W + La
        cur.commit(p)
    np.testing.assert_array_equal(np.concatenate(emitted), np.arange(T))     # every frame emitted exactly once, in order


@pytest.mark.parametrize("T,W,Ca,La", [(1601, 400, 40, 40), (1601, 400, 40, 0), (161, 40, 12, 8), (97, 8, 20, 3), (33, 5, 0, 0)])
def test_pipelined_stages_and_their_parity_buffers(T, W, Ca, La):
    """step_pipelined's three bodies: ENC / SD / D2 are double-buffered by step parity, and a stage of step k may only run
    once step k - 2 has left the buffer it writes - emulated by poisoning the buffers of parity k & 1 before step k."""
    cur = StreamCursor(W, Ca, La)
    n_max = W + La
    S, E = np.full(H + La + n_max + 64, -1), np.full(Ca + La + n_max + 64, -1)
    ENC = [np.full(n_max, -1), np.full(n_max, -1)]
    D2 = [np.full(H + n_max, -1), np.full(H + n_max, -1)]
    SD = [np.full(H + n_max, -1), np.full(H + n_max, -1)]
    emitted = []
    for n_new, last in _steps(T, W, La):
        p = cur.plan(n_new, last)
        par, h_dec, n_keep = p["par"], p["h_dec"], p["n_keep"]
        want = _expect(p["k"], p["e0"], p["e1"], W, Ca, last)
        ENC[par][:], SD[par][:] = -7, -7                               # whatever step k - 2 left there is dead by now
        # _front_body
        new = np.arange(p["e0"], p["e1"])
        S[p["n_tail"]:p["n_tail"] + n_new] = new
        np.testing.assert_array_equal(S[p["n_tail"] - p["h_enc"]:p["n_tail"] + n_new], want["enc"])
        ENC[par][:n_new] = new
        SD[par][:h_dec + n_keep] = S[p["dec_lo"]:p["dec_lo"] + h_dec + n_keep]
        if not last and p["spec_drop"] > 0:
            S[:p["n_tail"] + n_new - p["spec_drop"]] = S[p["spec_drop"]:p["n_tail"] + n_new].copy()
        # _mid_body (D2 of the OTHER parity is read for the history: it must still hold step k - 1's frames)
        keep_other = D2[par ^ 1].copy()
        D2[par][:] = -7
        E[p["n_ctx"]:p["n_ctx"] + n_new] = ENC[par][:n_new]
        x = E[:p["n_ctx"] + n_new].copy()
        np.testing.assert_array_equal(x, want["tscb"])
        if h_dec:
            D2[par][:h_dec] = keep_other[p["prev_len"] - h_dec:p["prev_len"]]
        D2[par][h_dec:h_dec + n_keep] = x[p["keep_lo"]:p["keep_lo"] + n_keep]
        if not last and p["enc_drop"] > 0:
            E[:p["n_ctx"] + n_new - p["enc_drop"]] = E[p["enc_drop"]:p["n_ctx"] + n_new].copy()
        # _dec_body
        np.testing.assert_array_equal(D2[par][:h_dec + n_keep], want["dec"])
        np.testing.assert_array_equal(SD[par][:h_dec + n_keep], want["dec"])
        emitted.append(D2[par][h_dec:h_dec + n_keep].copy())
        cur.commit(p)
    np.testing.assert_array_equal(np.concatenate(emitted), np.arange(T))


def test_a_step_with_the_wrong_number_of_frames_is_refused():
    cur = StreamCursor(40, 12, 8)
    with pytest.raises(ValueError):
        cur.plan(47, False)                                             # step 0 takes exactly 48 frames unless the clip ends
    with pytest.raises(ValueError):
        cur.plan(49, True)
    cur.commit(cur.plan(48, False))
    with pytest.raises(ValueError):
        cur.plan(41, False)
    p = cur.plan(3, True)                                               # the clip ends early: the last step emits what is left
    assert p["n_keep"] == 48 + 3 - 40 and p["last"]
