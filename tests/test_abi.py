"""The C-ABI library loads without a GPU and exports every symbol that
include/kzg_mi355x.h declares; the ctypes table binds exactly that set.
No compute calls here (CPU suite)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "kzg_mi355x.h")


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kzg_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    from kzg_snark_amd import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(built):
    out = subprocess.run(["nm", "-D", "--defined-only", built], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    missing = [s for s in header_symbols() if s not in exported]
    assert not missing, missing


def test_ctypes_table_matches_header(built):
    from kzg_snark_amd import _native
    assert sorted(_native.SIGNATURES) == header_symbols()
    L = _native.lib()
    assert _native.MISSING == []
    assert L.kzg_abi_version() == 1
    assert L.kzg_fp_limbs(0) == 4 and L.kzg_fp_limbs(1) == 6 and L.kzg_fp_limbs(7) == 0


def test_no_cpu_fallback(built):
    """Without a GPU a context cannot be created -- loudly.  (On a GPU box it can.)"""
    from kzg_snark_amd import _native
    try:
        ctx = _native.Context("bls12_381")
    except _native.NativeUnavailable as e:
        assert "no CPU fallback" in str(e)
    else:
        ctx.close()
    with pytest.raises(ValueError):
        _native.Context("secp256k1")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "kzg_snark_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "oracle/" not in text, f


def test_build_is_keyed_on_what_it_is_built_from(built, monkeypatch):
    """kzg_snark_amd/build.py rebuilds a translation unit when the hash of its source, the headers, the flags and the
    compiler changes -- not when a file time does -- and says which of the two it did."""
    from kzg_snark_amd import build
    stamp = build._read_stamp()
    assert stamp.get("source_hash") == build.source_hash()          # the library in the tree is built from these sources
    assert set(stamp.get("units", {})) == set(build.SOURCES)
    assert all(stamp["units"][u] == build.source_hash(u) for u in build.SOURCES)
    assert build.source_hash("ntt.hip") != build.source_hash("msm.hip")
    build.build(verbose=False)                                       # nothing changed: nothing compiled
    assert build.LAST["mode"] == "reused" and build.LAST["compiled"] == []
    os.utime(os.path.join(build.CSRC, "ntt.hip"))                    # a newer file time alone changes nothing
    build.build(verbose=False)
    assert build.LAST["mode"] == "reused"
    before = build.source_hash("ntt.hip")
    monkeypatch.setattr(build, "FLAGS", build.FLAGS + ["-DKZG_SOME_SWITCH=1"])
    assert build.source_hash("ntt.hip") != before                    # flags are part of the key


def test_header_is_plain_c_and_links_from_c(built, tmp_path):
    """The boundary is a C ABI: include/kzg_mi355x.h must compile as C99 (no C++ in the signatures) and a plain C
    program must link against the library and call it -- here the entry points that need no GPU (version, limb
    counts, the host-side sum of partial points with the generator added to itself and to its negative)."""
    src = tmp_path / "abi_probe.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "kzg_mi355x.h"
int main(void) {
  if (kzg_abi_version() != 1 || kzg_fp_limbs(KZG_CURVE_BN254) != 4 || kzg_fp_limbs(KZG_CURVE_BLS12_381) != 6) return 1;
  /* BN254 G1 generator (1, 2) and its negative (1, p - 2) */
  uint64_t xy[16] = {1, 0, 0, 0, 2, 0, 0, 0,
                     1, 0, 0, 0, 0x3c208c16d87cfd45ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  uint64_t out[8]; uint8_t inf = 9;
  if (kzg_g1_sum(KZG_CURVE_BN254, xy, NULL, 2, out, &inf) != KZG_OK || inf != 1) return 2;      /* G + (-G) = O */
  if (kzg_g1_sum(KZG_CURVE_BN254, xy, NULL, 1, out, &inf) != KZG_OK || inf != 0 || out[0] != 1 || out[4] != 2) return 3;
  uint64_t two[16]; memcpy(two, xy, 64); memcpy(two + 8, xy, 64);
  if (kzg_g1_sum(KZG_CURVE_BN254, two, NULL, 2, out, &inf) != KZG_OK || inf != 0) return 4;       /* G + G = 2G */
  if (out[0] != 0xd3c208c16d87cfd3ull || out[3] != 0x030644e72e131a02ull) return 5;               /* 2G.x (EIP-196) */
  if (kzg_ctx_create(7, 0, NULL) != KZG_ERR_ARG) return 6;
  puts("abi ok");
  return 0;
}
''')
    inc = os.path.join(ROOT, "include")
    libdir = os.path.dirname(built)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(inc, "kzg_mi355x.h")],
                   check=True)
    exe = tmp_path / "abi_probe"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lkzg_mi355x",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "abi ok" in out.stdout, (out.returncode, out.stdout, out.stderr[-500:])
