"""Are the kernels of two builds of libstrikeforce_amd.so the same machine code?

    python tools/isa_diff.py OLD.so NEW.so [substring of the mangled kernel name ...]

Extracts the gfx950 code object of each library (.hip_fatbin -> clang-offload-bundler), disassembles it
(llvm-objdump), cuts it per kernel symbol and compares the instruction text (addresses, encodings and branch-target
labels dropped).  SAME = identical text; SAME-OPS = every opcode occurs exactly as often as before (the register
allocator named or ordered something differently); DIFFERS = anything else.  Used to show that a change made for one kernel variant left the others byte for byte as they were
(round 4: the large-pool variant ZL next to the throughput kernels)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(so, out):
    fat = out + ".fatbin"
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out])
    return out


def kernels(co):
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True)
    out, name = {}, None
    for ln in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            name = subprocess.check_output(["c++filt", m.group(1)], text=True).strip()
            # (round 4 added a fourth template flag, `false` for every kernel that existed before)
            name = re.sub(r"(k_(?:reset|step|step_half)<\d+, \w+, \w+), false>", r"\1>", name)
            out[name] = []
            continue
        if name is None or not ln.strip():
            continue
        ins = ln.split("//")[0].strip()
        ins = re.sub(r"^[0-9a-f]+:\s*", "", ins)
        ins = re.sub(r"<[^>]+>", "<L>", ins)          # branch target labels
        ins = re.sub(r"\b(s_c?branch\w*|s_call\w*)\s+\S+", r"\1 <T>", ins)
        out[name].append(ins)
    return out


def main():
    old, new = sys.argv[1], sys.argv[2]
    pats = sys.argv[3:]
    with tempfile.TemporaryDirectory() as d:
        a = kernels(code_object(old, os.path.join(d, "a.co")))
        b = kernels(code_object(new, os.path.join(d, "b.co")))
    rc = 0
    for name in sorted(a):
        if pats and not any(p in name for p in pats):
            continue
        if name not in b:
            print("GONE     %6d ins  %s" % (len(a[name]), name))
            rc = 1
        elif a[name] == b[name]:
            print("SAME     %6d ins  %s" % (len(a[name]), name))
        elif sorted(i.split()[0] for i in a[name]) == sorted(i.split()[0] for i in b[name]):
            # the same instructions (every opcode as often as before), registers named or ordered differently
            print("SAME-OPS %6d ins  %s" % (len(a[name]), name))
        else:
            print("DIFFERS  %6d -> %d ins  %s" % (len(a[name]), len(b[name]), name))
            rc = 1
    for name in sorted(b):
        if name not in a and (not pats or any(p in name for p in pats)):
            print("NEW      %6d ins  %s" % (len(b[name]), name))
    return rc


if __name__ == "__main__":
    sys.exit(main())
