"""Voxel-grid PLY -> dense occupancy tensor (counterpart of the reference's build_sparse_occupancy.py).

Same command line and same output as the reference script (cuda_project_image_to_sparse_voxel/
build_sparse_occupancy.py:16-54): a ``torch.int32 [Z,Y,X]`` tensor whose occupied cells hold the
1-based PLY vertex index, saved with ``torch.save``.  Differences: the PLY is parsed here (the
``plyfile`` package is not needed), the per-vertex Python loop (BSO:45-46) is a single scatter, and
the functions are importable so that the aggregator can stay in one process.

Semantics kept:
  * coords = np.round((pts - origin) / voxel_size)  -- float32 arithmetic, round half to EVEN (BSO:32);
  * if any coordinate is negative the whole grid is shifted to start at 0 WITHOUT touching the
    grid origin (BSO:36-39) -- a warning is printed, exactly like the reference;
  * dims = max + 1 (BSO:41); on duplicates the LAST vertex wins (BSO:45-46).
"""
import argparse
import os
import re

import numpy as np
import torch

_PLY_TYPES = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
    "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
    "float": "f4", "float32": "f4", "double": "f8", "float64": "f8",
}


def extract_voxel_params(ply_path):
    """Header comments of the voxel-grid PLY (reference: aggregate_voxel_features_onthefly.py:65-92).

    Returns (voxel_size, grid_origin[3], grid_shape[3] or None, num_voxels_from_name or None).
    """
    voxel_size = grid_origin = grid_shape = num_voxels_from_name = None
    m = re.search(r"_(\d+)vox", os.path.basename(ply_path))
    if m:
        num_voxels_from_name = int(m.group(1))
    with open(ply_path, "rb") as f:
        for raw in f:
            try:
                line = raw.decode("ascii")
            except UnicodeDecodeError:
                break
            if "comment voxel_size" in line:
                voxel_size = float(line.split()[-1])
            if "comment grid_origin" in line:
                grid_origin = [float(x) for x in line.split()[-3:]]
            if "comment grid_shape" in line:
                grid_shape = [int(x) for x in line.split()[-3:]]
            if "end_header" in line:
                break
    if voxel_size is None or grid_origin is None:
        raise RuntimeError("Could not extract voxel_size or grid_origin from PLY header")
    return voxel_size, grid_origin, grid_shape, num_voxels_from_name


def read_voxel_ply(path):
    """Vertex positions of an ASCII or binary_little_endian PLY as float32 [N,3] (BSO:25-27)."""
    with open(path, "rb") as f:
        fmt, n_vertex, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise RuntimeError(f"{path}: no end_header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n_vertex = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise RuntimeError("list properties on vertices are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [p[0] for p in props]
        for k in "xyz":
            if k not in names:
                raise RuntimeError(f"{path}: vertex property {k} missing")
        if fmt == "ascii":
            rows = np.loadtxt(f, dtype=np.float64, max_rows=n_vertex, ndmin=2)
            pts = np.stack([rows[:, names.index(k)] for k in "xyz"], 1)
        elif fmt == "binary_little_endian":
            dt = np.dtype([(n, "<" + t) for n, t in props])
            rec = np.frombuffer(f.read(dt.itemsize * n_vertex), dtype=dt, count=n_vertex)
            pts = np.stack([rec[k] for k in "xyz"], 1)
        else:
            raise RuntimeError(f"{path}: unsupported PLY format {fmt}")
    return np.ascontiguousarray(pts, dtype=np.float32)


def voxel_coords(points, grid_origin, voxel_size):
    """Integer (x,y,z) cell of every point and the grid dims, BSO:30-41 (numpy float32, half-to-even)."""
    pts = np.asarray(points, dtype=np.float32)
    origin = np.array(grid_origin, dtype=np.float32)
    coords = np.round((pts - origin) / float(voxel_size)).astype(np.int64)
    min_coord = coords.min(axis=0)
    if not np.all(min_coord >= 0):
        print(f"Warning: negative min coords {min_coord}, will offset to zero")
        coords = coords - min_coord
    dims = coords.max(axis=0) + 1
    return coords, dims


def build_occupancy(points, grid_origin, voxel_size, device="cpu"):
    """Dense int32 [Z,Y,X] grid with occ[z,y,x] = vertex index + 1, last duplicate wins (BSO:44-46).

    On a CUDA device the whole of BSO:30-53 runs as two hand-written HIP kernels (vp_voxel_coords, vp_scatter_occupancy
    in csrc/vp_prep.h); the numpy / torch-CPU expression below is the command-line script's own host path (the
    reference script is a CPU program)."""
    if torch.device(device).type == "cuda":
        import voxproj_host
        pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(device)
        occ, lo = voxproj_host.build_occupancy_device(pts, grid_origin, voxel_size)
        if min(lo) < 0:
            print(f"Warning: negative min coords {np.array(lo)}, will offset to zero")
        return occ
    coords, dims = voxel_coords(points, grid_origin, voxel_size)
    dx, dy, dz = (int(v) for v in dims)
    lin = torch.from_numpy((coords[:, 2] * dy + coords[:, 1]) * dx + coords[:, 0]).to(device)
    ids = torch.arange(1, coords.shape[0] + 1, dtype=torch.int32, device=device)
    occ = torch.zeros(dz * dy * dx, dtype=torch.int32, device=device)
    # IDs grow with the vertex index, so "last writer wins" == "largest ID wins"
    occ.scatter_reduce_(0, lin, ids, reduce="amax", include_self=True)
    return occ.view(dz, dy, dx)


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--voxel_ply", required=True, help="path to voxel grid .ply")
    p.add_argument("--voxel_size", type=float, required=True, help="voxel size")
    p.add_argument("--grid_origin", nargs=3, type=float, default=[0, 0, 0], help="grid origin (x y z)")
    p.add_argument("--out_tensor", required=True, help="output path for occupancy tensor (.pt)")
    args = p.parse_args(argv)
    pts = read_voxel_ply(args.voxel_ply)
    print(f"Loaded {pts.shape[0]} points from PLY")
    occ = build_occupancy(pts, args.grid_origin, args.voxel_size)
    print(f"Occupancy tensor shape: {tuple(occ.shape)}  max ID: {int(occ.max())}  occupied: {int((occ > 0).sum())}")
    torch.save(occ, args.out_tensor)
    print(f"Saved occupancy tensor to {args.out_tensor}")


if __name__ == "__main__":
    main()
