#!/usr/bin/env python
"""Mechanical check of the cgo layer against the C ABI -- the image has no Go toolchain, so nothing else
ever compares `go/**/*.go` with `include/gosnark_hip.h`.

1. Every `C.gs_*( ... )` call in go/**/*.go is parsed (balanced parentheses, top-level commas) and compared with the prototype
   of the same name in include/gosnark_hip.h: the function must exist, the ARITY must match, and every argument must have the
   KIND of its parameter -- handle (gs_handle), pointer (T* / const T* / gs_handle*), or scalar with the same C type where the
   Go expression names one (C.size_t(..), C.int(..), C.uint64_t(..), C.uint32_t(..)).  Arguments whose kind cannot be derived
   from the expression or from a declaration in the enclosing function (`var x C.T`, `x := C.T(..)`) count as "unresolved"
   (reported, not failed).
2. Every header prototype must be bound by at least one Go call (the drop-in layer claims all entry points).
3. INTEGRATION.md pairs exported Go functions with plain-C drivers under tests/c/: every gs_* entry point reached from the Go
   functions named in a row (directly or through other functions of package gosnarkhip) must be called by that row's driver.

`python tools/check_go_abi.py` prints a summary and exits non-zero on a mismatch; tests/test_host_logic.py runs `check()`."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gosnark_hip.h")
GO_DIR = os.path.join(ROOT, "go")
C_TYPES = {"gs_handle", "gs_timing", "gs_memory", "gs_status"}
SCALAR_TYPES = {"size_t", "int", "uint64_t", "uint32_t", "unsigned", "uint8_t"}


# ---- header ----------------------------------------------------------------------------------------------------------
def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def parse_header(path=HEADER):
    """name -> (return type, [(kind, ctype)]) with kind in {'handle', 'pointer', 'scalar'}"""
    text = strip_c_comments(open(path).read())
    protos = {}
    for m in re.finditer(r"(?m)^\s*(int|void|const char\*|size_t)\s+(gs_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1), m.group(2), " ".join(m.group(3).split())
        plist = []
        if params and params != "void":
            for p in split_top(params):
                p = p.strip()
                if "*" in p or "[" in p:
                    base = re.sub(r"\bconst\b", "", p.split("*")[0].split("[")[0]).split()
                    # `uint64_t out[8]` has the name glued to the type list: the type is the first token(s) before the name
                    ctype = base[0] if base else "?"
                    plist.append(("pointer", ctype + "*"))
                else:
                    toks = re.sub(r"\bconst\b", "", p).split()
                    ctype = toks[0]
                    plist.append(("handle", "gs_handle") if ctype == "gs_handle" else ("scalar", ctype))
        protos[name] = (ret, plist)
    return protos


def split_top(s):
    """split at top-level commas (parentheses, brackets and braces nest)"""
    out, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    if "".join(cur).strip():
        out.append("".join(cur))
    return out


# ---- Go --------------------------------------------------------------------------------------------------------------
def strip_go_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: " " * len(m.group(0)), text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def go_files():
    for d, _, fs in os.walk(GO_DIR):
        for f in sorted(fs):
            if f.endswith(".go"):
                yield os.path.join(d, f)


def go_functions(text):
    """[(name, receiver type or None, body start, body end)] of the top-level funcs of a (comment-stripped) Go file"""
    funcs = []
    for m in re.finditer(r"(?m)^func\s*(\(\s*\w+\s+\*?(\w+)\s*\))?\s*(\w+)\s*\(", text):
        i = text.find("{", m.end())
        # the opening brace of the body is the first `{` at parenthesis depth 0 after the signature
        depth, j = 0, m.end() - 1
        while j < len(text):
            ch = text[j]
            if ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
            elif ch == "{" and depth == 0:
                i = j
                break
            j += 1
        depth, k = 0, i
        while k < len(text):
            if text[k] == "{":
                depth += 1
            elif text[k] == "}":
                depth -= 1
                if depth == 0:
                    break
            k += 1
        funcs.append((m.group(3), m.group(2), i, k))
    return funcs


def calls_in(text, lo, hi):
    """[(name, [args], position)] of the C.gs_* calls inside text[lo:hi]"""
    out = []
    for m in re.finditer(r"C\.(gs_[a-z0-9_]+)\s*\(", text[lo:hi]):
        if m.group(1) in C_TYPES:                      # a conversion C.gs_handle(x), not a call
            continue
        start = lo + m.end()
        depth, k = 1, start
        while k < len(text) and depth:
            if text[k] == "(":
                depth += 1
            elif text[k] == ")":
                depth -= 1
            k += 1
        out.append((m.group(1), [a.strip() for a in split_top(text[start:k - 1])], lo + m.start()))
    return out


def local_types(body):
    """identifier -> C type for `var x C.T`, `var x [n]C.T`, `x := C.T(...)`, `x, y := C.T(..), C.U(..)` inside a function body"""
    types = {}
    for m in re.finditer(r"\bvar\s+([\w,\s]+?)\s+(\[\w*\])?C\.(\w+)", body):
        for name in m.group(1).split(","):
            types[name.strip()] = ("array:" if m.group(2) else "") + m.group(3)
    # parameters of function literals and of the enclosing function: `func(out *C.uint64_t, inf *C.int) C.int { ... }`,
    # `func f(o *C.uint64_t, n C.size_t)` -- a `*C.T` parameter is a pointer, a `C.T` one a scalar / handle of that type
    for m in re.finditer(r"\bfunc\b[^()]*\(([^()]*)\)", body):
        pending = []
        for part in m.group(1).split(","):
            toks = part.split()
            if not toks:
                continue
            if len(toks) == 1:                 # `a, b *C.T`: the type follows a later name
                pending.append(toks[0])
                continue
            name, typ = toks[0], " ".join(toks[1:])
            mt = re.fullmatch(r"(\*?)C\.(\w+)", typ)
            for nm in pending + [name]:
                if mt and re.fullmatch(r"\w+", nm):
                    types[nm] = ("ptr:" if mt.group(1) else "") + mt.group(2)
            pending = []
    for m in re.finditer(r"(?m)([\w,\s]+?)\s*:=\s*(.+)$", body):
        names = [n.strip() for n in m.group(1).split(",")]
        vals = split_top(m.group(2))
        if len(names) == len(vals):
            for n, v in zip(names, vals):
                mv = re.match(r"\s*C\.(\w+)\(", v)
                if mv and re.fullmatch(r"\w+", n):
                    types[n] = mv.group(1)
                mp = re.match(r"\s*\(\*C\.(\w+)\)\(", v)      # o := (*C.uint64_t)(unsafe.Pointer(&out[0]))
                if mp and re.fullmatch(r"\w+", n):
                    types[n] = "ptr:" + mp.group(1)
    return types


def arg_kind(expr, types):
    """-> (kind, ctype or None); kind None = unresolved"""
    e = expr.strip()
    if e == "nil" or e.startswith("ptr(") or e.startswith("ptr32(") or e.startswith("ptrOrNil(") or e.startswith("(*C."):
        return "pointer", None
    m = re.match(r"C\.(\w+)\(", e)
    if m:
        t = m.group(1)
        return ("handle", "gs_handle") if t == "gs_handle" else ("scalar", t)
    if e.startswith("&"):
        return "pointer", None
    if re.fullmatch(r"\d+", e):                  # an untyped integer constant converts to whatever scalar the parameter is
        return "scalar", None
    if re.fullmatch(r"\w+", e) and e in types:
        t = types[e]
        if t.startswith("array:"):
            return None, None
        if t.startswith("ptr:"):
            return "pointer", None
        return ("handle", "gs_handle") if t == "gs_handle" else ("scalar", t)
    if e.startswith("unsafe.Pointer(") or e.startswith("(*") or e.startswith("handles(") or e.startswith("hptr("):
        return "pointer", None
    return None, None


def check():
    """-> (errors [str], stats dict)"""
    protos = parse_header()
    errors, bound, ncalls, unresolved = [], set(), 0, 0
    per_func_calls = {}        # (package dir, receiver, func name) -> set of gs_ names called directly
    per_func_refs = {}         # same key -> identifiers called in the body (for the transitive closure inside gosnarkhip)
    for path in go_files():
        text = strip_go_comments(open(path).read())
        rel = os.path.relpath(path, ROOT)
        pkg = os.path.basename(os.path.dirname(path))
        for fname, recv, lo, hi in go_functions(text):
            body = text[lo:hi]
            types = local_types(text[max(0, text.rfind("\nfunc", 0, lo)):hi])
            key = (pkg, recv, fname)
            per_func_calls.setdefault(key, set())
            per_func_refs[key] = set(re.findall(r"\b([A-Za-z_]\w*)\s*\(", body)) | set(re.findall(r"\.(\w+)\s*\(", body))
            for name, args, pos in calls_in(text, lo, hi):
                ncalls += 1
                line = text.count("\n", 0, pos) + 1
                where = "%s:%d (%s)" % (rel, line, fname)
                per_func_calls[key].add(name)
                if name not in protos:
                    errors.append("%s: C.%s is not declared in include/gosnark_hip.h" % (where, name))
                    continue
                bound.add(name)
                params = protos[name][1]
                if len(args) != len(params):
                    errors.append("%s: C.%s called with %d arguments, the header declares %d" % (where, name, len(args), len(params)))
                    continue
                for i, (a, (pk, pt)) in enumerate(zip(args, params)):
                    kind, ctype = arg_kind(a, types)
                    if kind is None:
                        unresolved += 1
                        continue
                    if kind != pk:
                        errors.append("%s: C.%s argument %d `%s` is a %s, the header wants a %s (%s)" % (where, name, i + 1, a[:40], kind, pk, pt))
                    elif kind == "scalar" and ctype and ctype in SCALAR_TYPES and pt in SCALAR_TYPES and ctype != pt:
                        errors.append("%s: C.%s argument %d `%s` is C.%s, the header wants %s" % (where, name, i + 1, a[:40], ctype, pt))
    for name in sorted(set(protos) - bound):
        errors.append("include/gosnark_hip.h declares %s, which no Go file calls" % name)
    pairs_checked, pair_errors = check_integration_pairs(per_func_calls, per_func_refs)
    errors += pair_errors
    return errors, {"prototypes": len(protos), "go_calls": ncalls, "bound": len(bound), "unresolved_arguments": unresolved,
                    "integration_rows_checked": pairs_checked}


# ---- Go functions <-> tests/c drivers ---------------------------------------------------------
HOUSEKEEPING = {"gs_last_error", "gs_version", "gs_set_device", "gs_get_device", "gs_device_count", "gs_handle_device"}


def closure(keys, per_func_calls, per_func_refs):
    """gs_* names reached from the given functions, following calls to other functions of package gosnarkhip by NAME"""
    by_name = {}
    for k in per_func_calls:
        if k[0] == "gosnarkhip":
            by_name.setdefault(k[2], []).append(k)
    seen, todo, names = set(), list(keys), set()
    while todo:
        k = todo.pop()
        if k in seen:
            continue
        seen.add(k)
        names |= per_func_calls.get(k, set())
        for ref in per_func_refs.get(k, ()):
            for k2 in by_name.get(ref, ()):
                if k2 not in seen and ref[:1].isupper() is False:       # only unexported helpers: exported ones are rows of their own
                    todo.append(k2)
    return names


def instance_helpers():
    """static helper functions of tests/c/instance.h: name -> body"""
    text = strip_c_comments(open(os.path.join(ROOT, "tests", "c", "instance.h")).read())
    out = {}
    text = text.replace("__attribute__((unused))", "")
    for m in re.finditer(r"static\s+[\w\s\*]+?\b(\w+)\s*\([^)]*\)\s*\{", text):
        depth, k = 0, m.end() - 1
        while k < len(text):
            if text[k] == "{":
                depth += 1
            elif text[k] == "}":
                depth -= 1
                if depth == 0:
                    break
            k += 1
        out[m.group(1)] = text[m.end():k]
    return out


def check_integration_pairs(per_func_calls, per_func_refs):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    errors, rows = [], 0
    for line in text.splitlines():
        if not line.startswith("| `") or "tests/c" in line and "driver" in line:
            continue
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(cells) < 3:
            continue
        drivers = re.findall(r"`(\w+\.c)`", cells[2])
        if not drivers:
            continue
        driver_names = set()
        for d in drivers:
            p = os.path.join(ROOT, "tests", "c", d)
            if not os.path.exists(p):
                errors.append("INTEGRATION.md names tests/c/%s, which does not exist" % d)
                continue
            src = strip_c_comments(open(p).read())
            driver_names |= set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", src))
            # helpers of tests/c/instance.h the driver calls (upload_groth_pk, ...) count with the entry points THEY call
            for hname, hbody in instance_helpers().items():
                if re.search(r"\b%s\s*\(" % hname, src):
                    driver_names |= set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", hbody))
        # Go functions of the row: gosnarkhip.Name and (*Type).Method
        keys = []
        for m in re.finditer(r"gosnarkhip\.(\w+)", cells[0]):
            keys += [k for k in per_func_calls if k[0] == "gosnarkhip" and k[1] is None and k[2] == m.group(1)]
        for m in re.finditer(r"\(\*?(\w+)\)\.(\w+(?:\s*/\s*\w+)*)", cells[0]):
            for meth in re.split(r"\s*/\s*", m.group(2)):
                keys += [k for k in per_func_calls if k[0] == "gosnarkhip" and k[1] == m.group(1) and k[2] == meth]
        if not keys:
            continue
        rows += 1
        need = closure(keys, per_func_calls, per_func_refs) - HOUSEKEEPING
        missing = sorted(need - driver_names)
        if missing:
            errors.append("INTEGRATION.md row `%s...`: driver(s) %s never call %s, which the row's Go functions do"
                          % (cells[0][:50], ", ".join(drivers), ", ".join(missing)))
    return rows, errors


if __name__ == "__main__":
    errs, stats = check()
    print("check_go_abi:", ", ".join("%s %s" % (v, k.replace("_", " ")) for k, v in stats.items()))
    for e in errs:
        print("  MISMATCH:", e)
    sys.exit(1 if errs else 0)
