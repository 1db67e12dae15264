"""Inputs of the front-end parity cases, shared by the golden generator (which runs
them through the reference's classes) and tests/test_frontend.py (which runs the product)."""
import hashlib

import numpy as np

from tests.golden.synth import synth_bits, synth_eti

UEP_BITRATES = (32, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384)
# (STL, TPL): every UEP bit rate x level, and EEP A / B levels at several bit rates
PUNCTURE_CASES = [(48, 0x22), (48, 0x20), (48, 0x21), (48, 0x23), (3, 0x21), (3, 0x20), (24, 0x24), (48, 0x25),
                  (96, 0x26), (12, 0x27), (144, 0x20), (6, 0x23)] + \
                 [(br * 3 // 8, lvl) for br in UEP_BITRATES for lvl in range(5)]
ETI_CASES = {
    # BASELINE config 1: Mode I, one 128 kbit/s sub-channel EEP 3-A, 48 frames = 12 transmission frames
    "cfg1": dict(nframes=48, mode=1, kw=dict()),
    # five sub-channels (UEP, EEP-B, EEP-A at 8 / 192 kbit/s, UEP 384 kbit/s), stream starting at FP = 5
    "multi": dict(nframes=43, mode=1, kw=dict(subchannels=((0, 48, 2), (100, 24, 0x25), (200, 3, 0x20),
                                                             (220, 72, 0x22), (400, 144, 0)), first_fct=245)),
    "mode2": dict(nframes=20, mode=2, kw=dict(subchannels=((0, 48, 0x22), (96, 12, 3)), mid=2, first_fct=3)),
    "mode3": dict(nframes=20, mode=3, kw=dict(subchannels=((0, 48, 0x22), (96, 12, 3)), mid=3, first_fct=3)),
    "mode4": dict(nframes=20, mode=4, kw=dict(subchannels=((0, 48, 0x22), (96, 12, 3)), mid=0, first_fct=3)),
}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_cases(fe):
    """fe: an odr-dabmod_amd.frontend.Frontend over the product or over the reference harness.
    Returns the dictionary of digests that tests/golden/golden.json stores under "frontend"."""
    g = {}
    g["prbs_head"] = bytes(fe.prbs(8)).hex()                       # 07 be 2e 64 12 9d a3 cf
    g["prbs_cif_padding"] = sha(fe.prbs(864 * 8))
    g["prbs_dispersal"] = sha(np.concatenate([fe.prbs(n, synth_bits(n, 7 + n)) for n in (96, 128, 24, 384, 1152)]))
    g["conv_8001"] = bytes(fe.conv_encode(np.array([0x80, 0x01], np.uint8))).hex()
    g["conv"] = sha(np.concatenate([fe.conv_encode(synth_bits(n, 70 + n)) for n in (96, 128, 24, 384)]))
    prof = []
    for stl in range(3, 200):
        for tpl in range(64):
            prof.append(repr((stl, tpl, fe.subchannel_profile(stl, tpl))))
    g["profiles"] = hashlib.sha256("\n".join(prof).encode()).hexdigest()
    g["profiles_valid"] = sum("None" not in p for p in prof)
    parts = [fe.puncture(synth_bits(4 * n + 3, 11 + mid), fic_mid=mid) for mid, n in ((1, 96), (3, 128))]
    for stl, tpl in PUNCTURE_CASES:
        if fe.subchannel_profile(stl, tpl) is not None:
            parts.append(fe.puncture(synth_bits(4 * stl * 8 + 3, stl * 64 + tpl), stl, tpl))
    g["puncture"] = sha(np.concatenate(parts))
    g["puncture_cases"] = len(parts)
    frames = np.stack([synth_bits(96 * 8, 50 + i) for i in range(40)])
    g["time_interleave"] = sha(fe.time_interleave(frames))
    for name, c in ETI_CASES.items():
        blocks = fe.eti_to_bits(synth_eti(c["nframes"], **c["kw"]), c["mode"])
        g["eti_" + name] = {"blocks": int(blocks.shape[0]), "block_bytes": int(blocks.shape[1]),
                            "sha256": sha(blocks), "first_block_sha256": sha(blocks[0])}
    return g
