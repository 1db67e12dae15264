#!/usr/bin/env python3
"""The hand-written prefetch of the PAIR kernels (tf_kernel.h: fetch_pair / park_pair) is invisible to the compiler's wait-count
bookkeeping: between the inline `global_load_dword vN, ...` and the inline `s_waitcnt vmcnt(0)` tied to vN, NOTHING may read or
write vN (a register copy or a spill placed there would read it before the load has landed).  This checks exactly that on the
device assembly of every instantiation that contains the pattern (no GPU needed); tests/test_cabi_cpu.py runs it.
usage: tools/check_pair_prefetch_asm.py            -> prints one line per kernel, exits 1 on a violation"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "odr-dabmod_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fvisibility=hidden", "-Xclang", "-target-feature", "-Xclang",
         "-load-store-opt", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include")]


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(line.rstrip("\n"))
            if "s_endpgm" in line:
                yield name, body
                name = None


def uses(reg, text):
    n = int(reg[1:])
    for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", text):
        if m.group(1) is not None:
            if int(m.group(1)) == n: return True
        elif int(m.group(2)) <= n <= int(m.group(3)): return True
    return False


def check(body):
    """returns (number of prefetch loads found, list of violations)"""
    found, bad = 0, []
    i = 0
    while i < len(body):
        if "#ASMSTART" in body[i] and i + 1 < len(body):
            m = re.match(r"\s*global_load_dword (v\d+), ", body[i + 1])
            if m:
                found += 1
                reg = m.group(1)
                j = i + 2
                waited = False
                while j < len(body):
                    t = re.sub(r";.*$", "", body[j]).strip()
                    if "#ASMSTART" in body[j] and j + 1 < len(body) and "s_waitcnt vmcnt(0)" in body[j + 1]:
                        waited = True
                        break
                    if t and not t.startswith(".") and not t.endswith(":") and uses(reg, t):
                        bad.append("%s touched before its wait: %s" % (reg, t))
                    if re.match(r"\s*(s_endpgm|s_branch)", body[j]) and False:
                        break
                    j += 1
                # (the load sits on the even-symbol path; the walk follows the text, which is the fall-through order of the loop)
                if not waited:
                    bad.append("%s: no tied wait found behind the load" % reg)
        i += 1
    return found, bad


def main():
    rc, total = 0, 0
    for logn, nt in ((11, 45),):
        out = tempfile.mktemp(suffix=".s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-DTF_LOGN=%d" % logn, "-DTF_NT=%d" % nt, "-o", out,
                              os.path.join(CSRC, "tf_inst.hip")], stderr=subprocess.DEVNULL)
        for name, body in kernels(out):
            found, bad = check(body)
            if not found: continue
            total += found
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            print("%-110s %d prefetch load(s): %s" % (dem[:110], found, "ok" if not bad else "; ".join(bad)))
            if bad: rc = 1
        os.unlink(out)
    print("%d hand-written prefetch loads checked" % total)
    if total == 0:
        print("no PAIR kernel found: the pattern this tool looks for has changed")
        rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
