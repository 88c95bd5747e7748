"""Readable kernel names for the profile summaries: demangle, drop the argument list, keep template arguments."""
import re
import subprocess
def pretty(k):
    """demangled kernel name without the argument list; template arguments kept (they tell the variants apart)"""
    if k.startswith("_Z"):
        try:
            d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
        except Exception:
            d = k
        if d.startswith("_Z"):        # binutils' c++filt does not know DF16_ / DF16b (_Float16 / __bf16): name + template arguments by hand
            import re
            m = re.match(r"_Z(\d+)", d)
            if m:
                n = int(m.group(1)); name = d[m.end():m.end() + n]; rest = d[m.end() + n:]
                args = []
                if rest.startswith("I"):
                    for tok in re.finditer(r"L([ib])(\d+)E|DF16(_|b)", rest[1:rest.find("EEv") + 1 if "EEv" in rest else len(rest)]):
                        if tok.group(1) == "i": args.append(tok.group(2))
                        elif tok.group(1) == "b": args.append("true" if tok.group(2) == "1" else "false")
                        else: args.append("_Float16" if tok.group(3) == "_" else "__bf16")
                d = name + ("<" + ", ".join(args) + ">" if args else "")
        k = d
    k = k.replace("void ", "").replace("(anonymous namespace)::", "")
    depth, out = 0, ""
    for ch in k:                      # cut at the first '(' that is outside template brackets
        if ch == "<": depth += 1
        if ch == ">": depth -= 1
        if ch == "(" and depth == 0: break
        out += ch
    return out[-96:]
