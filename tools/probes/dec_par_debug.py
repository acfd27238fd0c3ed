import sys, os, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmgan_amd import TSCNet
from cmgan_amd.synth import make_state_dict, synthetic_clips
m = TSCNet(64, 201).load_state_dict(make_state_dict(0)).eval()
eng = m.engine
wav = synthetic_clips(4, 8000, seed=1).cuda()
for name, fn in (("eager", lambda: eng.enhance(wav)), ("eager branched", lambda: eng.enhance_branched(wav, 2, 0)),
                 ("graph 1 branch", lambda: eng.enhance_graphed(wav, branches=1)), ("graph 1 branch replay", lambda: eng.enhance_graphed(wav, branches=1)),
                 ("graph 2 branches", lambda: eng.enhance_graphed(wav, branches=2)), ("graph 2 replay", lambda: eng.enhance_graphed(wav, branches=2))):
    print(name, flush=True)
    out = fn(); torch.cuda.synchronize()
    print("  ok", float(out.abs().max()), flush=True)
