import importlib, sys, hashlib, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
P = importlib.import_module("odr-dabmod_amd")
h = hashlib.sha256()
rs = np.random.RandomState(5)
for chunks in (0, 1, 3, 7, 11, 39, 77):
    for nf in (1, 2, 5, 16):
        for tii in (False, True):
            for gain in (2, 0, None):
                md = P.Modulator(mode=1, max_frames=nf, chunks_per_frame=chunks)
                stages = P.STAGE_FIR
                if gain is not None:
                    md.set_gain(gain, 1.0, 1 / 50000., 4.0); stages |= P.STAGE_GAIN
                if tii: md.set_tii(True, 3, 5, False)
                bits = np.frombuffer(rs.bytes(nf * 28800), np.uint8).reshape(nf, 28800)
                for _ in range(2):
                    y = md.chain(bits, stages)
                    h.update(y.tobytes())
                md.close()
print(h.hexdigest())
