"""CPU-side checks of the product: the C-ABI library loads and exports every declared symbol, the host logic of the
API mirror behaves like the reference's R code, and the product fails loudly without a GPU (no CPU fallback)."""
import os
import re
import sys

import numpy as np
import pytest

import __graft_entry__ as ge
from insider_amd import _lib, api, workloads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    ge.build()
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "insider_hip.h")).read()
    declared = set(re.findall(r"\b(insider_hip_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS)
    for s in declared:
        assert getattr(lib, s) is not None
    assert b"gfx950" in lib.insider_hip_version()


def test_product_never_imports_oracle():
    # only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/
    for dirpath, _, files in os.walk(os.path.join(ROOT, "insider_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("CPU oracle and the HIP kernels", ""), f


def test_fails_loudly_without_gpu(lib):
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    w = workloads.small()
    with pytest.raises(_lib.InsiderError) as e:
        api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    assert e.value.status == _lib.ERR_NO_DEVICE
    with pytest.raises(_lib.InsiderError):
        api.strong_coordinate_descent(None, None, np.zeros(3), 1.0, 0.5, np.eye(3), np.ones(3))


def test_argument_errors_do_not_exit():
    # the reference prints and exit(1)s on these (src/optimize.cpp:249-251,270-272)
    w = workloads.small()
    with pytest.raises(_lib.InsiderError, match="tuning should be either 0 or 1"):
        api.optimize(w.X, w.A0, w.C0, w.levels, None, w.M_train, w.M_test, 0, w.K, tuning=2)
    with pytest.raises(_lib.InsiderError, match="inc_continuous"):
        api.optimize(w.X, w.A0, w.C0, w.levels, None, w.M_train, w.M_test, 3, w.K)


def test_ratio_splitter_semantics():
    # R/utils.R:78-117
    rng = np.random.default_rng(0)
    d = rng.standard_normal((30, 20))
    d[rng.random(d.shape) < 0.1] = np.nan
    d[:, 3] = np.nan                      # a column that becomes all-zero is dropped
    out = api.ratio_splitter(d, ratio=0.2)
    keep = out["kept_columns"]
    assert not keep[3] and keep.sum() == 19
    tr, te, na = out["train_indicator"], out["test_indicator"], out["na_indicator"]
    assert not (tr & te).any() and not (tr & na).any() and not (te & na).any()
    assert (tr | te | na).all()
    n_existing = np.count_nonzero(~np.isnan(d))
    assert te.sum() <= int(np.floor(n_existing * 0.2))          # some may fall in the dropped column
    assert np.array_equal(out["testset"] != 0, te & (out["testset"] != 0))
    again = api.ratio_splitter(d, ratio=0.2)
    assert np.array_equal(again["test_indicator"], te)          # set.seed(123) analogue: deterministic


def test_insider_object_and_interaction_column():
    # R/insider.R:28-40: interaction indicator inserted as the SECOND column
    rng = np.random.default_rng(1)
    conf = workloads.cyclic_levels(24, (3, 2, 4))
    data = rng.standard_normal((24, 10))
    obj = api.insider(data, conf, interaction_idx=(1, 2))
    assert obj["confounder"].shape == (24, 4)
    assert np.array_equal(obj["confounder"][:, 0], conf[:, 0]) and np.array_equal(obj["confounder"][:, 2:], conf[:, 1:])
    inter = obj["confounder"][:, 1]
    pairs = {}
    for a, b, k in zip(conf[:, 0], conf[:, 1], inter):
        assert pairs.setdefault((a, b), k) == k
    assert len(set(pairs.values())) == len(pairs) == inter.max()
    assert obj["params"] == dict(global_tol=1e-9, sub_tol=1e-5, tuning_iter=30, max_iter=50000)
    with pytest.raises(ValueError, match="out of the range"):
        api.insider(data, conf, interaction_idx=(1, 9))
    with pytest.raises(ValueError, match="greater than or equal to 2"):
        api.insider(data, conf, interaction_idx=(1,))


def test_tune_argument_checks():
    obj = api.Insider(params=dict(global_tol=1e-9, sub_tol=1e-5, tuning_iter=3, max_iter=5))
    with pytest.raises(ValueError, match="TUNNING"):
        api.tune(obj, latent_dimension=None, lambda_=[1, 2])
    with pytest.raises(ValueError, match="TUNNING"):
        api.tune(obj, latent_dimension=np.array([5]), lambda_=1.0, alpha=0.1)


def test_workload_configs():
    w = workloads.make("c1")
    assert w.X.shape == (377, 5000) and list(w.n_levels) == [2, 16, 8, 107] and w.K == 23 and w.tuning == 0
    s = workloads.make(n=50, p=80, level_counts=(5, 2), K=3, f=0.1)
    assert s.M_test.sum() == int(np.floor(50 * 80 * 0.1)) and not (s.M_train & s.M_test).any()
    slab = workloads.make(n=50, p=3000, level_counts=(5, 2), K=3, f=0.1, gene_range=(1000, 2100))
    full = workloads.make(n=50, p=3000, level_counts=(5, 2), K=3, f=0.1)
    assert np.array_equal(slab.X, full.X[:, 1000:2100]) and np.array_equal(slab.M_train, full.M_train[:, 1000:2100])


def test_sweep_kernels_do_not_spill(tmp_path, lib):
    """The register-resident sweep kernels (insider_cd_reg.hpp) must not spill.

    Twice (KMAX = 22 in round 2, KMAX = 20 in round 3) a 128-VGPR build of the column-update kernel wrote its results to
    wrong addresses: the register allocator had put the spill STORE of a value live in all lanes inside an exec-masked
    region, the reload ran under the full mask, and the lanes that had been masked off used stale scratch.  The kernels are
    now written so that nothing needs spilling; this test disassembles the shipped code object and checks that
      * every instantiation up to KMAX = 30 (K <= 30 covers the BASELINE configurations) contains NO scratch instruction at
        all, the solve kernels (with their sweep loop), the evaluation kernels and the stand-alone batch solver alike;
      * KMAX = 32 (168 VGPRs for 128 matrix registers) keeps its sweep loop free of scratch traffic;
      * no kernel of the whole library stores a spill under a narrowed exec mask (tools/spill_scan.py)."""
    import re as _re
    import shutil
    import subprocess
    import sys as _sys
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    so = str(tmp_path / "libinsider_hip.so")
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.run([objdump, "--offloading", so], cwd=str(tmp_path), check=True, capture_output=True)
    co = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert co, "no gfx950 code object in the library"
    dis = subprocess.run([objdump, "-d", str(tmp_path / co[0])], check=True, capture_output=True, text=True).stdout
    _sys.path.insert(0, os.path.join(ROOT, "tools"))
    import spill_scan
    masked = {k: v for k, v in spill_scan.scan(dis).items() if v[1]}
    assert not masked, {k: v[2][:2] for k, v in masked.items()}
    # label -> address (kernels and the local labels of the sweep assembly, which split a kernel's listing)
    labels = {m.group(2): int(m.group(1), 16) for m in _re.finditer(r"^([0-9a-f]{16}) <([^>]+)>:", dis, _re.M)}
    funcs = _re.split(r"\n(?=[0-9a-f]{16} <_Z)", dis)
    loops = kernels = 0
    for fn in funcs:
        head = fn.split("\n", 1)[0]
        if "k_cd_cols_reg" not in head and "k_cd_batch_reg" not in head:
            continue
        kernels += 1
        kmax = int(_re.search(r"ILi[123]ELi(\d+)E", head).group(1))
        ins = []                                   # (address, instruction, rest of the line: branch targets live there)
        for line in fn.split("\n")[1:]:
            m = _re.search(r"^\s*(\S.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", line)
            if m:
                ins.append((int(m.group(2), 16), m.group(1), m.group(3)))
        scratch = [t for _a, t, _r in ins if t.startswith(("scratch_", "buffer_load", "buffer_store"))]
        if kmax <= 30:
            assert not scratch, (head, scratch[:4])
        # K <= 30 (round 5): the whole sweep LOOP is one asm statement — the successor list (absolute address pairs into
        # s[96 - 2 KMAX : 97]) is loaded at its entry and again in the exit block, for the next sweep; KMAX = 32 / three slots:
        # one statement per sweep, 32-bit offsets into s[64:97] / s[48:96], followed by the s_getpc_b64 that anchors the table
        pb = 96 - 2 * kmax
        first = f"s_load_dwordx16 s[{pb}:{pb + 15}]" if kmax <= 30 else ("s_load_dwordx16 s[48:63]" if kmax > 32 else "s_load_dwordx16 s[64:79]")
        sites = [i for i, (_, t, _r) in enumerate(ins) if t.startswith(first)]
        if "k_cd_cols_reg" in head and "ELb0E" in head:
            assert not sites and not any(t.startswith("s_setpc_b64") for _a, t, _r in ins), head   # the evaluation kernels have no sweep loop
            continue
        if kmax <= 30:
            assert len(sites) == 2, (head, len(sites))
            # one computed jump per code block and one into the first block, each on its own register pair; no address add
            jumps = [t for _a, t, _r in ins if t.startswith("s_setpc_b64 s[")]
            assert len(set(jumps)) == kmax + 1, (head, len(set(jumps)))
            # the column-update kernel also carries its blocks of two steps (a section of their own behind insider_cdpair_<KMAX>;
            # the disassembly lists them under the kernel in front of that label): one jump per block, 16 x 16 + W x W of them
            npair = 256 + (kmax - 16) ** 2 if "k_cd_cols_reg" in head else 0
            assert len(jumps) in (kmax + 1, kmax + 1 + npair), (head, len(jumps))
            assert not any(t.startswith("s_add_u32 vcc_lo") for _a, t, _r in ins), head
            # the loop body = from the table of blocks to the jump back into the first block (label Lgo): no scratch traffic, no
            # accumulation registers, no SGPR spill traffic (v_readlane / v_writelane) on the path of a sweep
            i0 = next(i for i, (_a, t, _r) in enumerate(ins) if t.startswith("s_lshl_b64 exec"))
            i1 = max(i for i, (_a, t, _r) in enumerate(ins[: sites[1] + 200]) if t.startswith(f"s_setpc_b64 s[{pb}:{pb + 1}]"))
            body = [t for _a, t, _r in ins[i0: i1 + 1]]
            assert len(body) > 9 * min(kmax, 30) and i1 > sites[1], (head, len(body))
            bad = [t for t in body if t.startswith(("scratch_", "buffer_load", "buffer_store", "v_accvgpr", "v_readlane", "v_writelane"))]
            assert not bad, (head, bad[:4])
            loops += 1
            continue
        assert len(sites) == 1, (head, len(sites))
        assert ins[sites[0] + (4 if kmax > 32 else 3)][1].startswith("s_getpc_b64 s[98:99]"), head
        a0 = ins[sites[0]][0]
        # the loop's back edge: the first branch after the sweep whose target lies at or shortly before the sweep's first load
        back = None
        for i in range(sites[0] + 1, len(ins)):
            addr, t, rest = ins[i]
            m = _re.search(r"<([^>+]+)(?:\+0x([0-9a-f]+))?>", rest) if _re.match(r"s_c?branch", t) else None
            target = None
            if m and m.group(1) in labels:
                target = labels[m.group(1)] + int(m.group(2) or "0", 16)
            elif t.startswith("s_setpc_b64 s[") and i >= 3 and ins[i - 3][1].startswith("s_getpc_b64"):
                # a relaxed (long) branch: s_getpc_b64 / s_add_u32 lo, lo, imm32 / s_addc_u32 hi, hi, -1|0 / s_setpc_b64 — the KMAX = 30
                # kernel's table of two-step blocks (62 KB, aligned to 64 KiB) lies inside its sweep loop
                ma = _re.match(r"s_add_u32 s\d+, s\d+, (0x[0-9a-f]+|-?\d+)", ins[i - 2][1])
                if ma:
                    imm = int(ma.group(1), 0)
                    imm -= (1 << 32) if imm >= (1 << 31) else 0
                    target = ins[i - 2][0] + imm
            if target is not None and target <= a0 and a0 - target < 256:
                back = (i, target)
                break
        assert back is not None, head
        body = [t for addr, t, _r in ins[: back[0] + 1] if addr >= back[1]]
        assert len(body) > 100, (head, len(body))                     # the whole sweep (code blocks + loss bookkeeping) is in it
        spills = [t for t in body if t.startswith(("scratch_", "buffer_load", "buffer_store"))]
        assert not spills, (head, spills[:4])
        if "k_cd_cols_reg" in head:   # ... nor parks values in accumulation registers (the three-slot kernels use up to 251 VGPRs)
            parked = [t for t in body if t.startswith("v_accvgpr")]
            assert not parked, (head, parked[:4])
        loops += 1
    # 9 register budgets x {solve, evaluate, stand-alone batch solver} + 4 three-slot budgets (32 < K <= 47) x {solve, batch solver}
    assert loops == 26 and kernels == 35, (loops, kernels)


def test_bench_gpus_n_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` from a plain shell (no torchrun environment) must start the N ranks itself, as a child
    process (torch.distributed.run, rendezvous on 127.0.0.1) and before anything touches the GPU."""
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_library_carries_the_hash_of_its_sources_and_a_stale_one_is_detected(tmp_path):
    """The benchmarked binary must provably be the committed source: the library reports the hash of the
    sources it was compiled from (insider_hip_version(): 'src:<sha16>' over csrc/, include/ and the flags), build() / _lib.load()
    compare it with the sources ON DISK by content — not by mtime — and rebuild on a mismatch.  Here: the in-tree library matches;
    a copy whose embedded hash is altered (what a binary built from other sources looks like) is reported stale; so is a missing one."""
    import shutil
    from insider_amd import _build, _lib
    assert not _build.needs_build(), (_build.library_sha(), _build.source_sha())
    assert _lib.library_source_sha() == _build.source_sha() == _build.library_sha()
    assert _lib.load().insider_hip_version().decode().endswith("src:" + _build.source_sha())
    stale = str(tmp_path / "libinsider_hip_stale.so")
    shutil.copy(_build.HIP_LIB, stale)
    blob = open(stale, "rb").read()
    tag = ("src:" + _build.source_sha()).encode()
    assert blob.count(tag) >= 1
    open(stale, "wb").write(blob.replace(tag, b"src:" + b"0" * 16))
    assert _build.library_sha(stale) == "0" * 16 and _build.needs_build(stale)
    assert _build.needs_build(str(tmp_path / "missing.so"))
    # a newer mtime on an up-to-date library changes nothing (the old rule would have trusted / distrusted it by time)
    os.utime(_build.HIP_LIB, None)
    assert not _build.needs_build()
