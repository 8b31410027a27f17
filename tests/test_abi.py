"""The C-ABI library loads and exports every symbol include/duckhts_amd.h declares (no compute calls)."""
import ctypes
import os
import re

import duckhts_amd
from conftest import ROOT


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dhts_[a-z0-9_]+|duckhts_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(duckhts_amd.LIB_PATH)
    names = declared_functions("duckhts_amd.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/duckhts_amd.h but not exported"
    assert sorted(duckhts_amd.EXPORTS) == names


def test_library_exports_nothing_the_headers_do_not_declare():
    """the reference hides everything but its entry point (CMakeLists.txt:92-95: C_VISIBILITY_PRESET hidden); here the dynamic symbol table
    is exactly what include/*.h declares -- no kernel stubs, pools or helpers (linker version script made from the headers by build())"""
    import subprocess
    import __graft_entry__ as ge
    declared = set(ge.declared_exports())
    out = subprocess.run(["nm", "-D", "--defined-only", duckhts_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported, "no dynamic symbols?"
    assert not (exported - declared), f"exported but not declared in include/: {sorted(exported - declared)}"
    diag_only = {"dhts_debug_diag", "dhts_debug_hw_diag", "dhts_debug_tr_diag"}      # defined by -DDHTS_DIAG / -DHW_DIAG / -DTR_DIAG builds only
    missing = {n for n in declared - exported - diag_only if n.startswith(("dhts_", "duckhts_", "register_", "duckdb_ext_api"))}
    assert not missing, f"declared but not exported: {sorted(missing)}"
    for n in ("duckhts_init_c_api", "register_read_bam_function", "register_read_bcf_function", "duckdb_ext_api"):
        assert n in exported


def test_abi_version_and_no_device_fails_loudly():
    L = duckhts_amd.lib()
    assert L.dhts_abi_version() == 1
    if L.dhts_device_count() == 0:
        # no GPU here: creating a context must fail (no CPU fallback)
        assert not L.dhts_create(0)
        try:
            duckhts_amd.Context(0)
        except duckhts_amd.DhtsError as e:
            assert "no CPU fallback" in str(e)
        else:
            raise AssertionError("Context() must raise without a device")
        assert b"no context" in L.dhts_error(None)


def test_numa_binding_of_the_calling_thread():
    """dhts_bind_thread_to_node binds the CALLING thread to the CPUs of a NUMA node (never to an empty set: the intersection with the CPUs
    the process may use), threads started afterwards inherit it; node -1 (unknown) and DHTS_NUMA=0 leave the thread alone"""
    import subprocess
    import sys
    import threading
    L = duckhts_amd.lib()
    assert L.dhts_bind_thread_to_node(-1) == 1
    res = {}

    def body():
        before = os.sched_getaffinity(0)
        rc = L.dhts_bind_thread_to_node(0)
        after = os.sched_getaffinity(0)
        child = {}
        t2 = threading.Thread(target=lambda: child.setdefault("aff", os.sched_getaffinity(0))); t2.start(); t2.join()
        res.update(rc=rc, before=before, after=after, child=child["aff"])
    t = threading.Thread(target=body); t.start(); t.join()
    assert res["rc"] in (0, 1)
    assert res["after"] and res["after"] <= res["before"] and res["child"] == res["after"]
    if res["rc"] == 0 and os.path.exists("/sys/devices/system/node/node0/cpulist"):
        node0 = set()
        for part in open("/sys/devices/system/node/node0/cpulist").read().strip().split(","):
            a, _, b = part.partition("-"); node0 |= set(range(int(a), int(b or a) + 1))
        assert res["after"] == (res["before"] & node0)
    assert os.sched_getaffinity(0) == res["before"] or threading.current_thread() is not threading.main_thread()      # the caller of the test is untouched
    code = "import sys; sys.path.insert(0, %r); import duckhts_amd; print(duckhts_amd.lib().dhts_bind_thread_to_node(0))" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, DHTS_NUMA="0"), timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "1", (r.stdout, r.stderr[-300:])
