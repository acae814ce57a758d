"""Repository policy checks (CPU): the product never reaches into oracle/ or the reference tree, and nothing that
runs on the GPU box reads /root/reference."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sources(*dirs, ext=(".py", ".hip", ".h", ".cpp")):
    for d in dirs:
        for base, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith(ext):
                    yield os.path.join(base, f)


def test_product_never_imports_the_oracle():
    bad = []
    for path in _sources("tgtc-style_amd", "tgtc_style_amd"):
        text = open(path).read()
        if re.search(r"^\s*(from|import)\s+oracle\b", text, re.M) or "oracle." in text.replace("oracle/", ""):
            bad.append(path)
    assert bad == []


def test_only_the_golden_generator_touches_the_reference():
    """Everything that may run on the GPU box (product, tests, bench, smoke, oracle) must not read /root/reference;
    tests/golden/gen_golden.py + ref_shim.py run in the build container only."""
    allowed = {os.path.join(ROOT, "tests", "golden", "gen_golden.py"), os.path.join(ROOT, "tests", "golden", "ref_shim.py"),
               os.path.abspath(__file__)}
    bad = []
    for path in list(_sources("tgtc-style_amd", "tgtc_style_amd", "oracle", "tests", "tools")) + \
            [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]:
        if path in allowed:
            continue
        text = open(path).read()
        if re.search(r"sys\.path[^\n]*reference|open\([^\n]*/root/reference|REFERENCE_ROOT", text):
            bad.append(path)
    assert bad == []


def test_no_cpu_fallback_in_the_product():
    """Operators must fail loudly without the HIP library / a GPU: no torch-eager restatement inside the package."""
    for path in _sources("tgtc-style_amd", ext=(".py",)):
        text = open(path).read()
        assert "F.linear" not in text and "torch.cumprod" not in text and "searchsorted" not in text, path
