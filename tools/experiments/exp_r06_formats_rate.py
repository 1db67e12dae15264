import importlib, json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
P = importlib.import_module("odr-dabmod_amd")
B = 16384
st = torch.cuda.Stream()
for fmt in ("s16", "u8", None):
    md = P.Modulator(mode=1, max_frames=B)
    md.set_gain(2, 1.0, 0.5, 4.0)
    md.set_output_format(fmt)
    with torch.cuda.stream(st):
        d_in = torch.randint(0, 256, (B, 28800), dtype=torch.uint8, device="cuda")
        out = torch.empty(B * 196608 * (4 if fmt == "s16" else 2 if fmt == "u8" else 8), dtype=torch.uint8, device="cuda")
        for _ in range(3): md.chain_dev(d_in, B, 3, out, stream=st.cuda_stream)
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(8): md.chain_dev(d_in, B, 3, out, stream=st.cuda_stream)
        e1.record(st); st.synchronize()
        ms = e0.elapsed_time(e1) / 8
    print(json.dumps({"format": fmt or "complexf", "frames_per_s": round(B / (ms * 1e-3)), "ms": round(ms, 3)}))
    md.close()
