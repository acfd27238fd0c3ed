"""What the carried-state streaming contract costs against the whole-clip pass, on the CPU oracle (test infrastructure:
oracle/stream_oracle.py vs oracle/cmgan_oracle.py): writes tests/golden/stream_cost_oracle.json, the bands of
tests/test_gpu_stream_config5.py.  ~6 min on 8 cores.    python tests/golden/make_stream_cost.py"""
import time, torch, sys, json, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import stream_oracle as S, cmgan_oracle as O
from oracle.weights import make_state_dict, synthetic_clips
from cmgan_amd import metrics as M
sd = make_state_dict(seed=0)
def cost(a, ref):
    a, ref = a.numpy().astype(np.float64), ref.numpy().astype(np.float64)
    d = a - ref
    ss = M.segmental_snr(ref, a, 16000)
    ss = ss[1] if isinstance(ss, tuple) else ss
    return dict(rel_max=float(np.abs(d).max()/np.abs(ref).max()), rel_rms=float(np.sqrt((d*d).mean()/(ref*ref).mean())),
                ssnr=float(np.mean(ss)), stoi=float(M.stoi(ref, a, 16000)))
out = {}
wav = synthetic_clips(1, 160000, seed=3)
whole = O.enhance_batch(sd, wav)[0]
for ca, la in ((40,40),(40,0)):
    s = S.enhance_stream(sd, wav, window=400, context=ca, lookahead=la)
    out[f"synthetic10s_w400_c{ca}_l{la}"] = cost(s, whole); print(out, flush=True)
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'tracks.npz'))
for name in ('a','b','silence'):
    pcm = torch.from_numpy(g['pcm_'+name].astype(np.float32)/32768.0)[None]
    L = pcm.size(1)//100*100
    pcm = pcm[:, :L]
    whole = O.enhance_batch(sd, pcm)[0]
    for w, ca, la in ((400,40,40),(100,40,40),(100,40,0)):
        s = S.enhance_stream(sd, pcm, window=w, context=ca, lookahead=la)
        out[f"track_{name}_w{w}_c{ca}_l{la}"] = cost(s, whole); print(name, w, ca, la, out[f"track_{name}_w{w}_c{ca}_l{la}"], flush=True)
json.dump(out, open(os.path.join(ROOT, 'tests', 'golden', 'stream_cost_oracle.json'), 'w'), indent=1)
