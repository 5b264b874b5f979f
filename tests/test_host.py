"""CPU tests of the host side: C-ABI library loads and exports every symbol include/fotg.h declares, host logic
(operating points, padding, derived parameters) matches the reference tables, no compute calls."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import flowonthego_amd as F
    hdr = open(os.path.join(ROOT, "include", "fotg.h")).read()
    declared = set(re.findall(r"\b(fotg_[a-z_0-9]+)\s*\(", hdr))
    declared.discard("fotg_ctx")
    bound = {s[0] for s in F._lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    L = F.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.fotg_version()


def test_operating_points_match_reference_table():
    # src/run_dense.cpp:168-209 / kroeger/run_dense.cpp:225-268; SURVEY.md section 8
    import flowonthego_amd as F
    from oracle import oracle as O
    for op_point in (1, 2, 3, 4):
        for w in (640, 1024, 1920, 3840):
            a, b = F.operating_point(op_point, w), O.op_point(op_point, w)
            assert (a.coarsest_scale, a.finest_scale, a.patch_size, a.grad_descent_iter, int(a.use_var_ref)) == \
                   (b.sc_f, b.sc_l, b.ps, b.max_iter, b.usetvref)
            assert abs(a.patch_stride - b.patove) < 1e-6
    op = F.operating_point(2, 1920)
    assert (op.coarsest_scale, op.finest_scale, op.steps, op.n_vals, op.n_scales) == (6, 4, 4, 64, 3)
    op = F.operating_point(4, 3840)
    assert (op.coarsest_scale, op.finest_scale, op.steps) == (7, 2, 3)
    assert F.AutoFirstScaleSelect(1024, 5, 8) == 5
    assert F.padded_size(1920, 1080, 6) == (1920, 1088, 0, 8)
    assert F.padded_size(1024, 436, 5) == (1024, 448, 0, 12)


def test_status_strings_and_bad_arguments():
    import ctypes as C
    import flowonthego_amd as F
    L = F.lib()
    assert L.fotg_strerror(0) == b"ok" and L.fotg_strerror(3) == b"batch larger than max_batch"
    assert L.fotg_op_point(2, -5, 1, F._lib.FotgParams()) == 1           # FOTG_ERR_ARG
    assert L.fotg_op_point(2, 640, 2, F._lib.FotgParams()) == 1
    assert L.fotg_padded_size(0, 10, 3, None, None, None, None) == 1
    with pytest.raises(F.FotgError):
        F._lib.check(1)
    # fotg_create's argument checks run before it touches a device (the sor_mode range check had been dropped, so 3 or -1
    # ran the lexicographic arithmetic without its buffers -- silently wrong flow on levels of more than 1024 rows)
    h = C.c_void_p()
    for field, bad in (("sor_mode", 3), ("sor_mode", -1), ("u8_color", 3), ("costfct", 3), ("noc", 2), ("tv_innerit", -1)):
        c = F._lib.FotgParams()
        assert L.fotg_op_point(2, 640, 1, c) == 0
        setattr(c, field, bad)
        assert L.fotg_create(C.byref(c), 640, 480, 0, 1, C.byref(h)) == 1, (field, bad)     # FOTG_ERR_ARG, whatever the box has for GPUs


def test_product_package_does_not_import_oracle():
    """the oracle is test infrastructure: nothing under flowonthego_amd/ may reference it"""
    pkg = os.path.join(ROOT, "flowonthego_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "dis_oracle" not in txt.replace("oracle/dis_oracle.c", ""), f


def test_only_tests_smoke_and_bench_touch_the_oracle():
    """tools/, examples/ and include/ never load the oracle; bench.py does so only inside cpu_baseline / its parity leg and
    __graft_entry__.py only inside build() (compiling the checker) and smoke()"""
    for d in ("tools", "examples", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, d)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".sh")):
                    txt = open(os.path.join(dp, f)).read()
                    assert "import oracle" not in txt and "from oracle" not in txt and "libdis_oracle" not in txt, os.path.join(dp, f)
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src.split("def cpu_baseline")[0]
    assert "from oracle" not in head and "import oracle" not in head           # no module-level import
    for line in src.splitlines():
        if "from oracle" in line or "import oracle" in line:
            assert line.startswith("    "), line                               # only inside functions (cpu_baseline, the parity leg)


def test_flo_writer_matches_reference_file_layout(tmp_path, alley_golden_flow):
    """write_flo == SaveFlowFile (kroeger/run_dense.cpp:16-57): tag, int32 w, int32 h, h*w*2 float32; the golden flow of the
    reference's own kroeger/flows/alley_0001.flo (tests/golden/alley_0001_flo.npz) survives a round trip bit for bit"""
    from flowonthego_amd.flo import read_flo, write_flo
    from oracle import oracle as O
    path = str(tmp_path / "a.flo")
    write_flo(path, alley_golden_flow)
    raw = open(path, "rb").read()
    h, w = alley_golden_flow.shape[:2]
    assert raw[:4] == b"PIEH" and np.frombuffer(raw[4:12], "<i4").tolist() == [w, h] and len(raw) == 12 + h * w * 8
    assert np.array_equal(read_flo(path), alley_golden_flow)
    assert np.array_equal(O.read_flo(path), alley_golden_flow)        # the oracle's reader agrees
    with pytest.raises(ValueError):
        write_flo(path, alley_golden_flow[..., 0])
    open(path, "wb").write(b"PIEX" + raw[4:])
    with pytest.raises(ValueError):
        read_flo(path)


def test_pfm_writer_matches_reference_file_layout(tmp_path):
    """write_pfm == SavePFMFile (kroeger/run_dense.cpp:60-81), restated here byte by byte: "Pf\\n%d %d\\n%f\\n" with scale -1,
    rows bottom-up, values negated"""
    import struct
    from flowonthego_amd.flo import read_pfm, write_pfm
    rng = np.random.default_rng(2)
    d = -np.abs(rng.standard_normal((7, 5))).astype(np.float32)
    path = str(tmp_path / "d.pfm")
    write_pfm(path, d[..., None])
    raw = open(path, "rb").read()
    exp = b"Pf\n5 7\n-1.000000\n"
    for y in range(6, -1, -1):
        for x in range(5):
            exp += struct.pack("<f", -d[y, x])
    assert raw == exp
    assert np.array_equal(read_pfm(path), d)
    with pytest.raises(ValueError):
        write_pfm(path, np.zeros((3, 3, 2), np.float32))


def _build_example(tmpdir, name="run_dense_min"):
    import subprocess
    exe = os.path.join(str(tmpdir), name)
    libdir = os.path.join(ROOT, "flowonthego_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", name + ".cpp"),
                           "-L" + libdir, "-lfotg", "-Wl,-rpath," + libdir, "-o", exe], stderr=subprocess.DEVNULL)
    return exe


def test_cpp_shim_reference_constructor_signatures(tmp_path):
    """a reference-side file that builds its own grids and refinement objects compiles against include/fotg/: the constructors
    have the reference's signatures -- PatGridClass(const img_params*, const opt_params*) (src/patchgrid.h:16, called like
    src/oflow.cpp:101), VarRefClass(const float*, const float*, const img_params*, const opt_params*, float*)
    (src/refine_variational.h:38-39, called like src/oflow.cpp:332) -- and dev_patch_state has the fields of src/patch.h:15-36"""
    import flowonthego_amd as F
    F.lib()
    exe = _build_example(tmp_path, "oflow_scale_loop")
    assert os.path.exists(exe)
    hdr = open(os.path.join(ROOT, "include", "fotg", "patchgrid.h")).read()
    assert "PatGridClass(const img_params *_i_params, const opt_params *_op)" in hdr
    hdr = open(os.path.join(ROOT, "include", "fotg", "refine_variational.h")).read()
    assert "VarRefClass(const float *_I0, const float *_I1, const img_params *_i_params, const opt_params *_op, float *flowout)" in hdr
    ref = open(os.path.join(ROOT, "include", "fotg", "patch.h")).read()
    for field in ("has_converged", "has_opt_started", "H00, H01, H11", "p_orgx, p_orgy", "p_curx, p_cury", "delta_px, delta_py",
                  "midpoint_curx, midpoint_cury", "midpoint_orgx, midpoint_orgy", "delta_p_sq_norm_init", "mares_old", "count", "invalid", "cost"):
        assert field in ref, field


def test_cpp_shim_example_builds(tmp_path):
    """the reference-side binding (include/fotg/{params,patchgrid,oflow}.h) compiles and links
    against libfotg.so as a run_dense-shaped C++ program"""
    import flowonthego_amd as F
    F.lib()
    exe = _build_example(tmp_path)
    assert os.path.exists(exe)


def test_cpp_pipeline_example_builds(tmp_path):
    """include/fotg/pipeline.h (several batches in flight, fotg_pipe_*) compiles and links as a video-loop program"""
    exe = _build_example(tmp_path, "video_pipeline")
    assert os.path.exists(exe)


def test_no_kernel_uses_scratch_memory():
    """the build keeps the compiler's per-kernel resource table (flowonthego_amd/libfotg.resusage.txt, written by the
    Makefile with -Rpass-analysis=kernel-resource-usage): no kernel of the product may have a private-memory segment --
    spills and dynamically indexed private arrays are HBM traffic the algorithm does not have (round 1: the four-patch LK
    kernel wrote 13x its algorithmic bytes this way)"""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "flowonthego_amd", "csrc")], stdout=subprocess.DEVNULL)
    txt = open(os.path.join(ROOT, "flowonthego_amd", "libfotg.resusage.txt")).read()
    names = re.findall(r"Function Name: (\S+)", txt)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", txt)]
    assert len(names) == len(scratch) and len(names) > 100
    bad = [(n, b) for n, b in zip(names, scratch) if b != 0]
    assert not bad, bad
    for must in ("lk_kernel", "vr_sor_tile_kernel", "pyr_base_kernel", "vr_sor_stream_kernel"):
        assert any(must in n for n in names), must


def test_launch_path_reads_no_environment():
    """every FOTG_* switch is read once at fotg_create; the only getenv in the library sits in the helper fotg_create uses,
    and the switches that change results (timing experiments) exist only in -DFOTG_DEBUG builds"""
    src = open(os.path.join(ROOT, "flowonthego_amd", "csrc", "fotg_capi.hip")).read()
    body = re.sub(r"#ifdef FOTG_DEBUG.*?#endif", "", src, flags=re.S)
    assert body.count("getenv(") == 1 and "static int env_int" in body
    for hdr in ("varref.hip.h", "varref_tiles.hip.h", "lk.hip.h", "pyramid.hip.h", "densify.hip.h", "varref_depth.hip.h", "common.h"):
        assert "getenv" not in open(os.path.join(ROOT, "flowonthego_amd", "csrc", hdr)).read()
    mk = [l for l in open(os.path.join(ROOT, "flowonthego_amd", "csrc", "Makefile")).read().splitlines() if not l.lstrip().startswith("#")]
    assert not any("FOTG_DEBUG" in l for l in mk)


# ---- bench.py as the N > 1 entry point (`--gpus` was parsed and never read) ----------------------
def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("fotg_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_gpus_flag_launches_one_rank_per_gpu():
    """`python bench.py --gpus N` without a launcher starts `python -m torch.distributed.run --nproc-per-node N ... bench.py
    --gpus N ...` as a child (stub here), relays its exit code, and refuses N > visible GPUs with a non-zero exit"""
    import types
    B = _load_bench()
    seen = {}

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)

    rc = B.self_launch(4, ["--gpus", "4", "--steps", "5"], device_count=lambda: 8, run=fake_run)
    assert rc == 7                                                     # the child's exit code is the launcher's
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    with pytest.raises(SystemExit) as e:
        B.self_launch(8, ["--gpus", "8"], device_count=lambda: 2, run=fake_run)
    assert e.value.code != 0


def test_bench_gpus_flag_cannot_print_a_line_for_another_n():
    """on this box (no GPU): `--gpus 2` exits non-zero with a clear message (0 GPUs visible), and a rank whose launcher started
    another number of ranks than --gpus (WORLD_SIZE=1, --gpus 2) exits non-zero BEFORE touching a GPU -- neither prints a
    metric line"""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr and '"metric"' not in r.stdout
        assert json.loads(r.stdout.strip().splitlines()[-1])["requested_gpus"] == 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and '"metric"' not in r.stdout


def test_bench_line_is_short(capsys, tmp_path, monkeypatch):
    """the driver keeps an 8 KB tail of stdout and round 5's 23 KB line could not be parsed.  bench.py's ONE stdout
    line is built by short_line() from the detailed result: here from round 5's real 23 KB result (profiles/r05_bench_line.json), also
    blown up to --gpus 8 (8 placements, per-rank times, scatter/gather totals) and with absurdly long notes -- always under
    SHORT_LINE_MAX bytes, always with the contract's keys, the roofline and the CPU baseline; the detail goes to a side file"""
    import json
    B = _load_bench()
    res = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line.json")))
    assert len(json.dumps(res)) > 20000                                # (the canned result really is the long one)
    res["config"].update(batch_per_gpu=64, sor="lexicographic")
    res["config_4k_op4_redblack"] = dict(res["config_4k_op4"], epe_vs_parity_mode_px={"mean": 0.05, "p99": 0.2, "max": 1.0})
    contract = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "one_batch_at_a_time", "roofline", "cpu_baseline", "whole_path_hbm_frac")

    def check(r):
        s = B.short_line(r)
        line = json.dumps(s, separators=(",", ":"))
        assert len(line) < B.SHORT_LINE_MAX <= 6000, len(line)
        for k in contract:
            assert k in s, k
        assert s["value"] == pytest.approx(r["value"], rel=1e-5) and s["ms_per_step"] == pytest.approx(r["ms_per_step"], rel=1e-5)
        assert set(s["config"]) >= {"workload", "global_batch", "pairs_in_flight"} and "model" not in s["config"]
        assert set(s["roofline"]) >= {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "ms_per_launch"}
        assert s["roofline"]["frac"] == pytest.approx(r["roofline"]["achieved"] / r["roofline"]["peak"], rel=1e-3)
        assert set(s["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample", "single_thread", "parity"}
        assert set(s["one_batch_at_a_time"]) == {"value", "ms_per_step"}
        for k in ("config_4k_op4", "config_4k_op4_fast_math", "config_4k_op4_redblack"):
            assert set(s[k]) >= {"ms_per_pair", "throughput"}
        return s, line

    s1, line1 = check(res)
    assert s1["n_gpus"] == 1 and s1["gpus_distinct"] == 1
    # --gpus 8
    r8 = json.loads(json.dumps(res))
    r8.update(n_gpus=8, rccl_ranks=8, ms_per_step_per_rank=[0.34 + 0.001 * k for k in range(8)],
              rank_placement=[{"rank": k, "local_rank": k, "pci_bus_id": "0000:%02x:00.0" % (5 + 8 * k)} for k in range(8)],
              scatter_gather={"scatter_plus_compute_ms": 12.5, "gather_ms": 0.4, "chunk_pairs": 16, "end_to_end_pairs_per_s": 39000.0,
                              "gathered_shape": [512, 68, 120, 2], "gathered_flows_match_single_context": True})
    s8, _ = check(r8)
    assert s8["rccl_ranks"] == 8 and s8["gpus_distinct"] == 8 and s8["ms_per_step_per_rank"] == {"min": 0.34, "max": 0.347}
    assert all(not isinstance(v, (list, dict)) for v in s8["scatter_gather"].values())          # scalars only
    # prose in the detail does not leak into the line
    rl = json.loads(json.dumps(r8))
    for k, v in rl.items():
        if isinstance(v, dict):
            v["note"] = "x" * 5000
    check(rl)
    # emit(): detail to the side file, the short line is the LAST (and only) stdout line
    monkeypatch.setattr(B, "ROOT", str(tmp_path))
    B.emit(json.loads(json.dumps(res)))
    out = capsys.readouterr()
    lines = out.out.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 6000 and json.loads(lines[0])["metric"] == res["metric"]
    det = json.load(open(tmp_path / "gpurun_out" / "bench_detail.json"))
    assert det["stage_ms"] == res["stage_ms"] and "rooflines" in det and "bench.py detail: " in out.err


def test_cpp_flow_writers_match_the_python_ones(tmp_path):
    """include/fotg/flowio.h: OFC::SaveFlowFile / OFC::SavePFMFile (kroeger/run_dense.cpp:16-81) write the bytes flowonthego_amd.flo
    writes (whose .flo layout is pinned against the reference's golden file in tests/test_oracle.py)"""
    import subprocess
    import numpy as np
    from flowonthego_amd import flo
    src = tmp_path / "w.cpp"
    src.write_text('#include "fotg/flowio.h"\n#include <vector>\nint main(int c, char **v) { const int w = 7, h = 5; std::vector<float> uv(2 * w * h), d(w * h);\n'
                   'for (int i = 0; i < 2 * w * h; ++i) uv[i] = 0.25f * i - 3.0f; for (int i = 0; i < w * h; ++i) d[i] = -0.5f * i;\n'
                   'return OFC::SaveFlowFile(uv.data(), w, h, v[1]) && OFC::SavePFMFile(d.data(), w, h, v[2]) && !OFC::SaveFlowFile(uv.data(), w, h, "/nonexistent/x.flo") ? 0 : 1; }\n')
    exe = tmp_path / "w"
    subprocess.check_call(["g++", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    subprocess.check_call([str(exe), str(tmp_path / "a.flo"), str(tmp_path / "a.pfm")])
    uv = (0.25 * np.arange(70, dtype=np.float32) - 3.0).astype(np.float32).reshape(5, 7, 2)
    d = (-0.5 * np.arange(35, dtype=np.float32)).reshape(5, 7)
    flo.write_flo(str(tmp_path / "b.flo"), uv)
    flo.write_pfm(str(tmp_path / "b.pfm"), d)
    assert (tmp_path / "a.flo").read_bytes() == (tmp_path / "b.flo").read_bytes()
    assert (tmp_path / "a.pfm").read_bytes() == (tmp_path / "b.pfm").read_bytes()
    assert np.array_equal(flo.read_flo(str(tmp_path / "a.flo")), uv) and np.array_equal(flo.read_pfm(str(tmp_path / "a.pfm")), d)
