"""Chunked producer->consumer pairs (CMGAN_CONV_CHUNKS / CMGAN_ATTN_CHUNKS, conformer.hip) must not change a bit:
   run the same batch in sub-processes with and without the knobs and compare the outputs."""
import os, subprocess, sys, hashlib
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from cmgan_amd import TSCNet
    from cmgan_amd.synth import make_state_dict, synthetic_clips
    m = TSCNet(64, 201, device="cuda:0")
    m.load_state_dict(make_state_dict(seed=0, num_features=201)).eval()
    wav = synthetic_clips(8, 32000, seed=7).to("cuda:0")
    out = m.engine.enhance(wav)
    torch.cuda.synchronize()
    print("SHA", hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest())
    sys.exit(0)
ref = None
for env in ({}, {"CMGAN_CONV_CHUNKS": "4"}, {"CMGAN_ATTN_CHUNKS": "4"}, {"CMGAN_CONV_CHUNKS": "3", "CMGAN_ATTN_CHUNKS": "2"}):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
    sha = [l for l in r.stdout.splitlines() if l.startswith("SHA")]
    print(env, sha[0] if sha else r.stderr[-400:])
    if sha:
        ref = ref or sha[0]
        assert sha[0] == ref, "chunked form differs"
print("chunked forms bit-identical")
