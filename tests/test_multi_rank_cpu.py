"""N > 1 path on CPU: world_size-2 gloo processes shard the views (r::G) and run the REAL scene-combination code of the
entry point -- VoxelFeatureAggregator.all_reduce (all-reduce and reduce-to-rank-0 forms) and .result() -- on per-rank
partial {sum, count, views} tensors.  Only the projection itself is stood in for (there is no GPU here; tests may use the
oracle for that): each rank writes the oracle's per-view sums into its aggregator's state exactly where the projector's
gather would have accumulated them.  The combined result must equal the single-rank result: counts and view counts
exactly, sums to fp32 rounding, and the output rows (voxel order, coordinates, fp16 means) of result() likewise."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "3d-semantic-segmentation_amd")


def _scene():
    from synthetic_scene import make_features_np, make_scene
    s = make_scene(2000, 7, 40, 24, seed=61, room=(5.0, 4.0, 2.4))
    return s, make_features_np(7, 24, 40, 8, seed=61)


def _project(oracle, s, feats, views, count, sums, nviews):
    for v in views:
        c1 = np.zeros_like(count)
        oracle.project_features(feats[None, v:v + 1], s.occ[None].astype(np.int64), s.c2w[v].reshape(-1), s.intr[None],
                                s.opts(), s.grid_origin, s.voxel_size, c1, sums)
        count += c1
        nviews += c1 > 0


def _filled_aggregator(oracle, s, feats, views):
    """A fast-mode VoxelFeatureAggregator (CPU tensors) whose state holds what projecting `views` leaves there."""
    from aggregate_voxel_features_onthefly import VoxelFeatureAggregator
    agg = VoxelFeatureAggregator(torch.from_numpy(s.occ), s.grid_origin.astype(np.float64), s.voxel_size, 8, "fast", "cpu")
    n_rows = s.n_vox + 1
    assert agg.n_rows == n_rows
    count, sums, nviews = np.zeros(n_rows, np.int32), np.zeros((n_rows, 8), np.float32), np.zeros(n_rows, np.int32)
    _project(oracle, s, feats, views, count, sums, nviews)
    agg.sum32.copy_(torch.from_numpy(sums))
    agg.count.copy_(torch.from_numpy(count))
    agg.views.copy_(torch.from_numpy(nviews))
    agg.n_seen = len(list(views))
    return agg


def _final_call_stand_in(oracle, agg, s, feats, log):
    """Stands in for the projector inside add_final_views (there is no GPU here): `feats` carries VIEW INDICES; the call
    accumulates the oracle's sums of those views for the voxel IDs of the row range now set on the aggregator's workspace --
    what vp_project_features / VP_FLAG_GATHER_ONLY leave in {sum32, count, views}."""
    import voxproj_host as vh
    n_rows = agg.n_rows

    def project(view_ids, vmi, intr, gather_only=False):
        lo = agg.ws.options.get(vh.VP_OPT_ROW_BEGIN, 0)
        hi = agg.ws.options.get(vh.VP_OPT_ROW_END, n_rows)
        log.append((bool(gather_only), lo, hi))
        count, sums, nviews = np.zeros(n_rows, np.int32), np.zeros((n_rows, 8), np.float32), np.zeros(n_rows, np.int32)
        _project(oracle, s, feats, [int(v) for v in view_ids], count, sums, nviews)
        agg.sum32[lo:hi] += torch.from_numpy(sums[lo:hi])
        agg.count[lo:hi] += torch.from_numpy(count[lo:hi])
        agg.views[lo:hi] += torch.from_numpy(nviews[lo:hi])

    agg._stage = lambda f, c, i: (f, c, i, None)
    agg._project_fast = project


def _worker(rank, world, port, out_path):
    for p in (ROOT, PKG):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from view_sharding import views_of_rank
    s, feats = _scene()
    mine = views_of_rank(s.n_views, rank, world)
    assert mine == list(range(rank, 7, world))
    res = {}
    # every collective the aggregator issues, by op and element count (round 5: the integers travel in ONE tensor)
    issued = []
    real_all_reduce, real_reduce = dist.all_reduce, dist.reduce

    def counting_all_reduce(t, *a, **k):
        issued.append(("all_reduce", str(t.dtype), t.numel()))
        return real_all_reduce(t, *a, **k)

    def counting_reduce(t, *a, **k):
        issued.append(("reduce", str(t.dtype), t.numel()))
        return real_reduce(t, *a, **k)

    dist.all_reduce, dist.reduce = counting_all_reduce, counting_reduce
    n_rows = s.n_vox + 1
    n_ints = 2 * ((n_rows + 63) & ~63) + 64
    for name, dst in (("all", None), ("root", 0)):
        agg = _filled_aggregator(oracle, s, feats, mine)
        assert agg.count.data_ptr() == agg._ints.data_ptr() and agg._ints.numel() == n_ints      # views into the one tensor
        del issued[:]
        agg.all_reduce(dst=dst)                                # the entry point's own combination step
        if dst is None:                                        # the sums + ONE integer tensor {pixel counts, view counts, views seen}
            assert issued == [("all_reduce", "torch.float32", n_rows * 8), ("all_reduce", "torch.int32", n_ints)], issued
        else:                                                  # reduce to the root, then the root's view total to everyone
            assert issued == [("reduce", "torch.float32", n_rows * 8), ("reduce", "torch.int32", n_ints),
                              ("all_reduce", "torch.int32", 1)], issued
        assert agg.n_seen == 7
        if rank == 0 or dst is None:
            r = agg.result()
            res[name] = dict(sums=agg.sum32.numpy().copy(), count=agg.count.numpy().copy(), nviews=agg.views.numpy().copy(),
                             avg=r["avg_feats"].numpy(), coords=r["voxel_coords"].numpy(), xyz=r["xyz"].numpy(),
                             hit_count=r["hit_count"].numpy())
    # The split form the entry point ships (add_final_views -> view_sharding.project_final_call_and_reduce): all views but the
    # rank's last two as ordinary calls, the last two as the final call, cut at split_point(n_rows); then the unsplit arm.
    from view_sharding import split_point
    import voxproj_host as vh
    for name, split in (("split", True), ("unsplit", False)):
        agg = _filled_aggregator(oracle, s, feats, mine[:-2])
        log = []
        _final_call_stand_in(oracle, agg, s, feats, log)
        del issued[:]
        h = agg.add_final_views(torch.tensor(mine[-2:]), torch.zeros(2, 4, 4), torch.zeros(4), dst=None, split=split)
        assert h == (split_point(n_rows) if split else 0) and 0 <= h < n_rows and h % 64 == 0
        if split:                                              # the sums in two pieces (the first under the second gather) + the integers
            assert issued == [("all_reduce", "torch.float32", h * 8), ("all_reduce", "torch.float32", (n_rows - h) * 8),
                              ("all_reduce", "torch.int32", n_ints)], issued
        else:
            assert issued == [("all_reduce", "torch.float32", n_rows * 8), ("all_reduce", "torch.int32", n_ints)], issued
        assert log == ([(False, 0, h), (True, h, n_rows)] if split else [(False, 0, n_rows)]), log
        assert vh.VP_OPT_ROW_BEGIN not in agg.ws.options and vh.VP_OPT_ROW_END not in agg.ws.options     # range reset
        assert agg.n_seen == 7
        r = agg.result()
        res[name] = dict(sums=agg.sum32.numpy().copy(), count=agg.count.numpy().copy(), nviews=agg.views.numpy().copy(),
                         avg=r["avg_feats"].numpy(), coords=r["voxel_coords"].numpy(), xyz=r["xyz"].numpy(),
                         hit_count=r["hit_count"].numpy())
    assert res["split"]["sums"].tobytes() == res["unsplit"]["sums"].tobytes()      # the cut changes no bit of the reduced scene
    if rank == 1:                                              # after an all-reduce every rank holds the scene
        np.savez(out_path + ".rank1.npz", **{k: v for k, v in res["all"].items()})
    if rank == 0:
        np.savez(out_path, **{f"{n}_{k}": v for n, d in res.items() for k, v in d.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_view_sharding_equals_single_rank(tmp_path, oracle_mod):
    out_path = str(tmp_path / "r.npz")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out_path), nprocs=2, join=True)
    got = np.load(out_path)
    rank1 = np.load(out_path + ".rank1.npz")
    s, feats = _scene()
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    single = _filled_aggregator(oracle_mod, s, feats, range(7))
    r = single.result()
    count, sums, nviews = single.count.numpy(), single.sum32.numpy(), single.views.numpy()
    assert count.sum() > 5000
    for name in ("all", "root", "split", "unsplit"):
        assert np.array_equal(got[f"{name}_count"], count) and np.array_equal(got[f"{name}_nviews"], nviews)
        np.testing.assert_allclose(got[f"{name}_sums"], sums, rtol=1e-5, atol=1e-6)
        # the files' rows: same voxels in the same order, same coordinates, fp16 means equal up to the fp32 summation order
        assert np.array_equal(got[f"{name}_coords"], r["voxel_coords"].numpy())
        assert np.array_equal(got[f"{name}_hit_count"], r["hit_count"].numpy())
        assert got[f"{name}_xyz"].tobytes() == r["xyz"].numpy().tobytes()
        np.testing.assert_allclose(got[f"{name}_avg"].astype(np.float32), r["avg_feats"].numpy().astype(np.float32), rtol=2e-3, atol=1e-4)
    assert np.array_equal(rank1["count"], count) and np.array_equal(rank1["coords"], r["voxel_coords"].numpy())
    assert rank1["sums"].tobytes() == got["all_sums"].tobytes()          # an all-reduce leaves the same bits on every rank


def test_views_of_rank_partition():
    from view_sharding import views_of_rank
    for n, g in ((300, 8), (7, 2), (3, 4), (216, 3)):
        parts = [views_of_rank(n, r, g) for r in range(g)]
        assert sorted(v for p in parts for v in p) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
