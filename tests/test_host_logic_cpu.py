"""CPU: host-side logic added in round 6 -- the handle table that replaced the tensor-attribute hints, the stream plan of a
rank, and the launch-geometry queries of the C library that Python sizes buffers from (no compute calls: no GPU here)."""
import gc
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_handle_table_follows_object_identity_and_version():
    from minsu3d_amd.backend import _HandleTable
    t = _HandleTable()
    a = torch.zeros(8, dtype=torch.int32)
    assert t.get(a) is None and t.get(a, -1) == -1
    t.put(a, 1)
    assert t.get(a) == 1
    assert t.get(a.clone()) is None and t.get(a[:4]) is None and t.get(a.view(2, 4)) is None   # copies / views: not vouched for
    a.add_(1)                                   # an in-place write voids the entry
    assert t.get(a) is None
    t.put(a, 0)
    assert t.get(a, -1) == 0                    # a stored 0 is a value, not "absent"
    del a
    gc.collect()
    assert not t._entries                       # the entry dies with the tensor (ids are recycled, objects are not)
    b = torch.zeros(8, dtype=torch.int32)
    assert t.get(b) is None


def test_stream_plan_keeps_a_data_parallel_rank_inside_four_hardware_queues(monkeypatch):
    from minsu3d_amd.parallel import stream_plan
    for k in ("MS3D_PREFETCH_STREAM", "MS3D_FORCE_PG", "MS3D_WGRAD_STREAM", "MS3D_EARLY_HEADS", "MS3D_PREFETCH_COORDS"):
        monkeypatch.delenv(k, raising=False)
    one = stream_plan(1)
    assert one["prefetch_stream"] == "own" and one["compute_streams"] == 3 and one["collective_streams"] == 0
    eight = stream_plan(8)
    assert eight["prefetch_stream"] == "side" and eight["compute_streams"] == 2 and eight["collective_streams"] != 0
    monkeypatch.setenv("MS3D_FORCE_PG", "1")
    assert stream_plan(1)["prefetch_stream"] == "side"
    monkeypatch.setenv("MS3D_PREFETCH_STREAM", "own")
    assert stream_plan(8)["compute_streams"] == 3
    monkeypatch.setenv("MS3D_PREFETCH_COORDS", "0")
    assert stream_plan(8)["prefetch_stream"] == "off" and stream_plan(8)["compute_streams"] == 2


def test_geometry_queries_are_consistent():
    """the functions Python sizes buffers from agree with each other for the shapes of the three models (pure host code)"""
    from minsu3d_amd import _lib
    lib = _lib.lib()
    # 32 -> 32 on a full-resolution table: 64-row list by default, 32-row list for dense tables; nothing else changes
    for V in (196_000, 417_000):
        assert lib.ms3d_spconv_pairlist_rows(V, 27, 32, 32) == 64
        assert lib.ms3d_spconv_pairlist_rows_dense(V, 27, 32, 32) == 32
        for cin, cout in ((16, 16), (16, 32), (32, 16)):
            assert lib.ms3d_spconv_pairlist_rows_dense(V, 27, cin, cout) == lib.ms3d_spconv_pairlist_rows(V, 27, cin, cout) == 64
        # both column blocks per wave: one partial row per workgroup instead of one per (workgroup, slice)
        assert lib.ms3d_spconv_partial_blocks(V, 27, 32, 32, 32) * 2 == lib.ms3d_spconv_partial_blocks(V, 27, 32, 32, 64)
        assert lib.ms3d_spconv_partial_blocks(V, 27, 32, 32, 1) == lib.ms3d_spconv_partial_blocks(V, 27, 32, 32, 64)
    assert lib.ms3d_spconv_pairlist_rows_dense(2_600, 27, 32, 32) == 0       # below the pair-list threshold: no list at all
    assert lib.ms3d_kmap_pairlist_rows_of(None) == 0
    # the weight-stationary route: by default the K = 27 layers beyond 256 channels on 64+ tiles; a unit per (row part, slice)
    ws = lib.ms3d_spconv_partial_blocks(2_600, 27, 320, 160, 0)
    tiles = lib.ms3d_spconv_partial_blocks(2_600, 27, 160, 160, 0)
    assert 0 < ws < tiles
    assert lib.ms3d_spconv_partial_blocks(500, 27, 384, 192, 0) > 0


def test_weight_stationary_switches_are_read_from_the_environment():
    code = ("import sys; sys.path.insert(0, %r); from minsu3d_amd import _lib; l = _lib.lib(); "
            "print(l.ms3d_spconv_partial_blocks(2600, 27, 160, 160, 0), l.ms3d_spconv_partial_blocks(2600, 27, 320, 160, 0))" % ROOT)
    out = {}
    for name, env in (("default", {}), ("all", {"MS3D_WS_ALL": "1"}), ("off", {"MS3D_WS_MAX_TILES": "0"})):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = tuple(int(x) for x in r.stdout.split())
    assert out["default"][0] == out["off"][0] and out["default"][1] != out["off"][1]      # only the wide layer is routed
    assert out["all"][0] != out["off"][0] and out["all"][1] == out["default"][1]
