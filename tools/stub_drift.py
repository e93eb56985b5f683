#!/usr/bin/env python3
"""Does tests/velox_api_stub/velox_stub.h still say what the reference's headers say?

The shim (shim/*.cpp) is compiled against the stub because Velox itself cannot be built in this image;
the stub is only worth something while its names ARE the reference's. This script makes drift visible:
for every section of the stub ("// ---- exec/ ----", "// ---- type/Type.h ----" ...) it collects the
classes / structs / enums the section declares and the member and free functions declared in them, and
looks each identifier up in the reference headers the section names (a directory = every header in it).
It prints what it could not find and exits 1 if there is any; run where /root/reference exists
(tests/test_shim.py::test_stub_names_exist_in_the_reference does, and is skipped elsewhere).

    python tools/stub_drift.py [/root/reference/velox]
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "velox_api_stub", "velox_stub.h")

# identifiers that are the stub's own machinery (test behaviour, helpers), not the reference's API
OWN = {
    "detail", "format", "formatInto", "main", "operator", "if", "for", "while", "switch", "return", "sizeof", "static_cast",
    "reinterpret_cast", "const_cast", "dynamic_cast", "decltype", "defined", "assert", "alignas", "catch", "throw", "new",
    "delete", "noexcept", "explicit", "override", "final", "template", "typename", "std", "folly",
}


# Names the stub declares on purpose without a counterpart: what they stand in for.
STAND_INS = {
    "CpuOperatorStandIn": "any CPU operator of the reference a test plan keeps (Values, TableScan ...)",
    "mutableOperators": "test access to Driver::operators_ (DriverAdapter::adapt gets it through DriverFactory)",
    "mutableQueryConfig": "test access to QueryCtx's config",
    "statsCopy": "Operator::stats(bool) hands out a copy under the reference's lock",
    "mutableBlocks": "the two-line accessor INTEGRATION.md asks Velox's SplitBlockBloomFilter wrapper for",
    "numBlocksHeld": "same accessor pair",
    "vx355_bloom_num_blocks": "the C ABI's own function, declared where the stub's Bloom filter needs it",
    "Simple": "template parameter name",
    "FutureState": "shared state behind the stand-in for folly::SemiFuture / folly::Promise (folly is not in the reference tree)",
    "isReady": "folly::SemiFuture::isReady (folly is not in the reference tree)",
    "ConstantVectorBase": "untyped base of the stub's ConstantVector<T> (the shim only uses BaseVector's interface on it)",
    "DictionaryVectorBase": "untyped base of the stub's DictionaryVector<T>",
    "AndOfFilters": "what Filter::mergeWith yields for two filters on one column (the reference builds typed merged filters)",
}


def sections(text):
    """[(header spec, body text)] per '// ---- spec ----' comment."""
    marks = [(m.start(), m.group(1).strip()) for m in re.finditer(r"^// ---- (.*?) -{2,}\s*$", text, re.M)]
    out = []
    for i, (at, spec) in enumerate(marks):
        end = marks[i + 1][0] if i + 1 < len(marks) else len(text)
        out.append((spec, text[at:end]))
    return out


def headers_of(spec, ref):
    """Reference files a section's comment names: 'a/b.h, c/d.h (note)', 'exec/' = the directory, and siblings
    named without their directory ('vector/BaseVector.h, FlatVector.h')."""
    spec = re.sub(r"\(.*?\)", "", spec)
    files, last_dir = [], ""
    for part in [p.strip() for p in spec.split(",") if p.strip()]:
        part = re.sub(r":\d+(-\d+)?$", "", part)
        if "/" in part:
            last_dir = part.rsplit("/", 1)[0]
        else:
            part = os.path.join(last_dir, part)
        path = os.path.join(ref, part)
        if part.endswith("/") or os.path.isdir(path):
            for base, _, names in os.walk(path):
                if "/tests" in base or "/benchmarks" in base:
                    continue
                files += [os.path.join(base, n) for n in names if n.endswith(".h")]
        elif os.path.exists(path):
            files.append(path)
    return files


def declared(body):
    """Type names and function names a section declares."""
    names = set()
    body = re.sub(r"//.*", "", body)
    body = re.sub(r'"(\\.|[^"\\])*"', '""', body)
    for m in re.finditer(r"\b(?:class|struct|enum class|enum)\s+([A-Za-z_]\w*)\s*(?:final\s*)?[:{]", body):
        names.add(m.group(1))
    # function declarations: a return type or qualifier before the name, '(' after it, not a call inside a body
    for m in re.finditer(r"^\s*(?:virtual\s+|static\s+|inline\s+|explicit\s+|constexpr\s+|friend\s+)*"
                         r"(?:[\w:<>,\*&\s]+?[\s\*&])?([A-Za-z_]\w*)\s*\([^;{}]*\)\s*(?:const\s*)?(?:noexcept\s*)?"
                         r"(?:override\s*)?(?:final\s*)?(?:=\s*0\s*)?[;{]", body, re.M):
        name = m.group(1)
        if name not in OWN and not name.startswith("VELOX_") and not name.isupper():
            names.add(name)
    return names


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/velox"
    if not os.path.isdir(ref):
        print(f"stub_drift: {ref} does not exist here - nothing to compare with")
        return 2
    text = open(STUB).read()
    everywhere = None   # identifiers of every header of the reference (built on first need)
    missing, elsewhere, checked = [], [], 0
    for spec, body in sections(text):
        files = headers_of(spec, ref)
        if not files:
            print(f"stub_drift: section '{spec}' names no reference header that exists")
            missing.append((spec, "<section headers>"))
            continue
        ref_text = "\n".join(open(f, errors="replace").read() for f in files)
        words = set(re.findall(r"[A-Za-z_]\w*", ref_text))
        for name in sorted(declared(body)):
            if name in STAND_INS:
                continue
            checked += 1
            if name in words:
                continue
            if everywhere is None:
                everywhere = set()
                for base, _, names in os.walk(ref):
                    if "/tests" in base or "/benchmarks" in base or "/experimental" in base:
                        continue
                    for n in names:
                        if n.endswith(".h"):
                            everywhere |= set(re.findall(r"[A-Za-z_]\w*", open(os.path.join(base, n), errors="replace").read()))
            (elsewhere if name in everywhere else missing).append((spec, name))
    for spec, name in elsewhere:
        print(f"stub_drift: note: '{name}' (section '{spec}') lives in another header of the reference")
    for spec, name in missing:
        print(f"stub_drift: '{name}' (section '{spec}') is in NO header of the reference")
    print(f"stub_drift: {checked} declared names checked, {len(elsewhere)} in other headers, {len(missing)} not found; "
          f"{len(STAND_INS)} names are the stub's own stand-ins")
    return 1 if missing else 0


if __name__ == "__main__":
    sys.exit(main())
