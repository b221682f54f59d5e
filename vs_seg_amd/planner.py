"""Host-side planning for the implicit-GEMM kernels: lattice classes, tap tables, weight pack maps, tile/LDS choices.

Pure numpy — no GPU needed — so every index rule is unit-tested on CPU (tests/test_planner.py) by running
`simulate_igemm`, a literal restatement of the kernel's indexing, against the convolution definitions.

Lattice form of every convolution-like op of the network (ref:params/networks/blocks/convolutions.py:114-146 and the
autograd of those ops):

    out[q*os + oo][n] = sum_t sum_c in[q*is + off_t][c] * W[t][c][n]

  kind           in      out     classes   per-dim rule (k = kernel, s = stride, p = (k-1)//2)
  conv_fwd       X       Y       1         os=1 oo=0 is=s, taps off=d-p                      (d = 0..k-1)
  convT_dgrad    dY      dX      1         same as conv_fwd
  convT_fwd      X       Y       prod(s)   os=s oo=par is=1, taps off=e with d=par+p-s*e in [0,k)
  conv_dgrad     dY      dX      prod(s)   same as convT_fwd
"""
from __future__ import annotations

import dataclasses
import itertools
from dataclasses import dataclass, field
from typing import Optional,  List, Sequence, Tuple

import numpy as np

LDS_LIMIT = 160 * 1024


@dataclass
class LatticeClass:
    os: Tuple[int, int, int]
    oo: Tuple[int, int, int]
    is_: Tuple[int, int, int]
    taps: List[Tuple[Tuple[int, int, int], Tuple[int, int, int]]]  # (input offset, kernel index) per tap


def same_pad(kernel):
    return tuple((k - 1) // 2 for k in kernel)


def lattice_classes(kind: str, kernel: Sequence[int], stride: Sequence[int]) -> List[LatticeClass]:
    pad = same_pad(kernel)
    per_dim = []
    for k, s, p in zip(kernel, stride, pad):
        opts = []
        if kind in ("conv_fwd", "convT_dgrad"):
            opts.append((1, 0, s, [(d - p, d) for d in range(k)]))
        elif kind in ("convT_fwd", "conv_dgrad"):
            for par in range(s):
                taps = []
                for e in range(-k, k + 1):
                    d = par + p - s * e
                    if 0 <= d < k:
                        taps.append((e, d))
                assert taps, "a parity class without taps would leave output voxels unwritten"
                opts.append((s, par, 1, sorted(taps)))
        else:
            raise ValueError(kind)
        per_dim.append(opts)
    classes = []
    for ox, oy, oz in itertools.product(*per_dim):
        taps = [((a[0], b[0], c[0]), (a[1], b[1], c[1])) for a in ox[3] for b in oy[3] for c in oz[3]]
        classes.append(LatticeClass((ox[0], oy[0], oz[0]), (ox[1], oy[1], oz[1]), (ox[2], oy[2], oz[2]), taps))
    return classes


def weight_flat_index(kind: str, wshape: Sequence[int], c: np.ndarray, n: np.ndarray, d: np.ndarray) -> np.ndarray:
    """Flat index into the torch-layout weight for GEMM-K channel c, GEMM-N channel n, flat kernel index d."""
    K = int(np.prod(wshape[2:]))
    if kind == "conv_fwd":  # W[co=n][ci=c]
        return (n * wshape[1] + c) * K + d
    if kind == "convT_fwd":  # W[ci=c][co=n]
        return (c * wshape[1] + n) * K + d
    if kind == "conv_dgrad":  # K = co, N = ci: W[co=c][ci=n]
        return (c * wshape[1] + n) * K + d
    if kind == "convT_dgrad":  # K = co, N = ci: W[ci=n][co=c]
        return (n * wshape[1] + c) * K + d
    raise ValueError(kind)


def gemm_dims(kind: str, wshape: Sequence[int]) -> Tuple[int, int]:
    """(K channels, N channels) of the implicit GEMM."""
    if kind == "conv_fwd":
        return wshape[1], wshape[0]
    if kind == "convT_fwd":
        return wshape[0], wshape[1]
    if kind == "conv_dgrad":
        return wshape[0], wshape[1]
    if kind == "convT_dgrad":
        return wshape[1], wshape[0]
    raise ValueError(kind)


def round_up(a, b):
    return (a + b - 1) // b * b


@dataclass
class IgemmPlan:
    kind: str
    cls: LatticeClass
    q: Tuple[int, int, int]
    kc: int  # padded K channels (multiple of 8) as laid out in the input tensor
    nc: int  # true N channels
    kreal: int  # true K channels
    tile: Tuple[int, int, int]
    mtw: int
    nt: int
    nsplit: int
    ck: int
    nchunks: int
    ksteps: int
    lds: int
    depth: int = 1
    pack_map: np.ndarray = field(repr=False, default=None)  # int32, -1 = zero
    res_tiles: int = 0  # marching plans: 16-channel tiles of a 1x1x1 convolution of the same input riding along (march_res_plans) ...
    pack_map_res: np.ndarray = field(repr=False, default=None)  # ... and their gather map
    classes: Optional[List[LatticeClass]] = field(repr=False, default=None)  # class-split plan: workgroup row s computes lattice class s (cls = their union, oo = 0)

    def class_taps(self, s: int) -> List[int]:
        """Class-split plan: index into the union tap table (cls.taps) of each tap of class s."""
        offs = [t[0] for t in self.cls.taps]
        return [offs.index(off) for off, _ in self.classes[s].taps]

    @property
    def ntaps(self):
        return len(self.cls.taps)


HALO_MAX = 8 * 256 * 16  # bytes of one halo chunk the kernel can fetch per stage (PMAX pieces of 16 B per thread)


def igemm_halo_bytes(tile, is_, taps, ck, es):
    halo = 1
    for a in range(3):
        offs = [t[0][a] for t in taps]
        halo *= (tile[a] - 1) * is_[a] + (max(offs) - min(offs) + 1)
    return halo * ck * es


def igemm_lds_bytes(tile, is_, taps, ck, ksteps, nt, mtw, es, nchunks=1, aux_es=4, depth=1):
    """Mirror of igemm_prepare() in csrc/igemm.hip: tap table | epilogue constants | weights (x2 when streamed) | 2 halo buffers | aux | coordinate tables."""
    w = ksteps * nt * 64 * 8 * es
    nbuf = max(depth, 0) + 1  # depth -1: no prefetch, single buffer
    aux = nbuf * round_up(64 * mtw * nt * 16 * aux_es + 64 * mtw * 4, 1024) if (aux_es and 64 * mtw * nt * aux_es <= (12 if mtw == 8 else 8) * 256) else 0  # DMA-prefetched residual / accumulate tile (AMAX pieces per thread) + gate floats
    hb = igemm_halo_bytes(tile, is_, taps, ck, es)
    tables = ((0 if nt >= 3 else 2 * ((hb // 16 + 255) // 256)) + (64 * mtw + 255) // 256) * 1024  # per-thread tables of the DMA pieces (nt <= 2 only) + tile-voxel coordinates
    return round_up(ksteps * 16, 16) + 3 * nt * 16 * 4 + w * (nbuf if nchunks > 1 else 1) + nbuf * round_up(hb, 1024) + aux + tables  # ring buffers padded to whole 1 KiB DMA instructions


# ---- streaming kernel (csrc/sconv.hip): depth -2 ------------------------------------------------------------------------
def stream_mt(kc, ntaps) -> int:
    """M-tiles per wave of the streaming kernel (sc_mt() in csrc/sconv.hip): tile = (2 * mt, 8, 4)."""
    return 2 if (kc >= 64 and ntaps == 9) else 4


def stream_tile(kc, ntaps):
    return (2 * stream_mt(kc, ntaps), 8, 4)


STREAM_SHAPES = {(8, 1, 9), (8, 2, 9), (16, 1, 9), (16, 2, 9), (16, 4, 9), (32, 1, 9), (32, 2, 9), (32, 4, 9),
                 (16, 1, 1), (16, 2, 1), (32, 1, 1), (32, 2, 1), (32, 4, 1), (64, 2, 1), (64, 4, 1), (64, 2, 9),
                 (32, 3, 1), (48, 2, 1), (48, 3, 1), (48, 6, 1), (96, 3, 1), (96, 6, 1)}  # (input channels, 16-channel output tiles, taps) instantiated by sconv.hip
_TAPS_3x3x1 = [(t // 3 - 1, t % 3 - 1, 0) for t in range(9)]


def stream_eligible(cls: "LatticeClass", q, kc, nreal, es) -> bool:
    """Mirror of sc_find() in csrc/sconv.hip for what a plan decides (the tensors' layout is checked by the library, loudly)."""
    offs = [tuple(t[0]) for t in cls.taps]
    if es != 2 or tuple(cls.is_) != (1, 1, 1) or tuple(cls.os) != (1, 1, 1) or tuple(cls.oo) != (0, 0, 0):
        return False
    if offs != _TAPS_3x3x1 and offs != [(0, 0, 0)]:
        return False
    if any(v % t for v, t in zip(q, stream_tile(kc, len(offs)))):
        return False
    return (kc, (nreal + 15) // 16, len(offs)) in STREAM_SHAPES


def stream_lds_bytes(kc, nt, ntaps):
    g = kc // 8
    r = {9: 2, 4: 1}.get(ntaps, 0)  # halo voxels added per axis (x, y)
    tile = stream_tile(kc, ntaps)
    pieces = (tile[0] + r) * (tile[1] + r) * tile[2] * g
    return ((ntaps * g + 3) // 4) * nt * 1024 + ((pieces + 255) // 256) * 4096 + 5 * nt * 16 * 4 + 16


def stream_plan(kind, wshape, cls, q, es, kc, nreal, kreal) -> Optional["IgemmPlan"]:
    """The depth -2 candidate: the compile-time-geometry streaming kernel on the launches it covers (stride-1 3x3x1 / 1x1x1 bf16,
    the instantiated (input channels, output tiles, taps) of STREAM_SHAPES — up to 64 channels either side, up to 96 for the 1x1x1 launches of level 2 —, extents
    divisible by the 8x8x4 tile)."""
    if not stream_eligible(cls, q, kc, nreal, es):
        return None
    nt = (nreal + 15) // 16
    ntaps = len(cls.taps)
    return IgemmPlan(kind, cls, tuple(q), kc, nreal, kreal, stream_tile(kc, ntaps), stream_mt(kc, ntaps), nt, 1, kc, 1, (ntaps * (kc // 8) + 3) // 4, stream_lds_bytes(kc, nt, ntaps), -2)


# ---- marching streaming kernel (csrc/mconv.hip): depth -5 ------------------------------------------------------------------
# (input channels, 16-channel output tiles, TZ, M-tiles per wave) instantiated by mconv.hip; rows per workgroup TYB = 64 * mt / tz
MARCH_SHAPES = {(16, 1, 4, 2), (16, 1, 8, 4), (16, 2, 4, 2), (16, 2, 8, 4), (32, 1, 4, 2), (32, 2, 4, 2), (8, 1, 8, 4), (8, 2, 8, 4), (8, 1, 8, 8), (8, 2, 8, 8), (8, 1, 4, 8), (8, 2, 4, 8), (8, 1, 4, 4), (8, 2, 4, 4), (16, 1, 4, 8), (16, 1, 4, 4), (16, 2, 4, 8), (16, 2, 4, 4), (16, 2, 8, 8), (16, 1, 8, 8),
                (32, 1, 2, 4), (32, 1, 4, 4), (32, 1, 2, 2), (32, 2, 4, 4), (32, 2, 2, 4), (32, 2, 2, 2), (32, 4, 4, 4), (32, 4, 2, 2), (32, 4, 4, 2),
                (64, 2, 2, 2), (64, 2, 2, 1), (64, 1, 2, 2), (64, 1, 2, 1)}
# ... of which these also exist with the packed weights in registers instead of LDS (depth -6; csrc/mconv.hip WREG, the MC_W entries)
MARCH_WREG_SHAPES = {(16, 2, 4, 2), (16, 2, 8, 4), (32, 1, 4, 2), (32, 2, 4, 2), (32, 1, 2, 4), (32, 1, 4, 4), (32, 1, 2, 2), (32, 2, 4, 4), (32, 2, 2, 4), (32, 2, 2, 2), (32, 4, 2, 2), (32, 4, 4, 2),
                     (64, 2, 2, 2), (64, 2, 2, 1), (64, 1, 2, 2), (64, 1, 2, 1)}
MARCH_RING = 4
MARCH_DEPTHS = (-5, -6)


def is_march(pl) -> bool:
    return pl.depth in MARCH_DEPTHS


def march_lds_bytes(kc, nt, tz, mt, wreg=False):
    g = kc // 8
    rows = mt * 4 * (16 // tz) + 2
    return max((0 if wreg else ((9 * g + 3) // 4) * nt * 1024) + MARCH_RING * round_up(rows * tz * g * 16, 256) + 5 * nt * 16 * 4 + 16, 4 * 2 * nt * 16 * 4)


def march_plans(kind, wshape, cls, q, es, kc, nreal, kreal, n=1) -> List["IgemmPlan"]:
    """The depth -5 candidates: the marching kernel (every input voxel fetched once: a workgroup owns a column of `tyb` rows x `tz` slices and
    walks along x with a ring of planes in LDS) on the stride-1 3x3x1 bf16 launches it is instantiated for.  tile = (x steps per workgroup,
    rows per workgroup, tz); x is cut into segments so that the launch has about two (or one) rounds of 512 resident workgroups."""
    offs = [tuple(t[0]) for t in cls.taps]
    if es != 2 or tuple(cls.is_) != (1, 1, 1) or tuple(cls.os) != (1, 1, 1) or tuple(cls.oo) != (0, 0, 0) or offs != _TAPS_3x3x1 or kc != round_up(kreal, 8):
        return []
    nt = (nreal + 15) // 16
    out = []
    for (c, t, tz, mt) in sorted(MARCH_SHAPES):
        tyb = 64 * mt // tz
        if c != kc or t != nt or q[1] % tyb or q[2] % tz or march_lds_bytes(kc, nt, tz, mt) > 158 * 1024:
            continue
        cols = n * (q[1] // tyb) * (q[2] // tz)
        for target in (512, 1024):
            nxs = max(1, min(q[0] // 8, -(-target // cols)))
            lx = -(-q[0] // nxs)
            pl = IgemmPlan(kind, cls, tuple(q), kc, nreal, kreal, (lx, tyb, tz), mt, nt, 1, kc, 1, (9 * (kc // 8) + 3) // 4, march_lds_bytes(kc, nt, tz, mt), -5)
            if not any(o.tile == pl.tile and o.mtw == pl.mtw for o in out):
                out.append(pl)
                if (c, t, tz, mt) in MARCH_WREG_SHAPES:  # the same launch with the packed weights in registers (LDS bandwidth back to the operand reads)
                    out.append(dataclasses.replace(pl, depth=-6, lds=march_lds_bytes(kc, nt, tz, mt, True)))
    return out


# ---- gathering marching kernel (csrc/gconv.hip): depth -9 ------------------------------------------------------------------
# (input channels, 16-channel output tiles, tz, M-tiles per wave): gc_table of csrc/gconv.hip
GATHER_SHAPES = {(16, 1, 4, 2), (16, 1, 8, 4), (16, 1, 4, 4), (16, 2, 4, 2), (16, 2, 8, 4), (16, 2, 4, 4), (32, 2, 2, 1), (32, 2, 4, 2), (32, 3, 2, 1), (32, 3, 4, 2)}


def gather_lds_bytes(kc, nt, tz, mt) -> int:
    """Mirror of gc_lds() in csrc/gconv.hip: packed weights | ring of six fine planes (two half planes of tyb + 1 rows each, padded to 256 bytes; the plane to whole 1 KiB DMA rows) | epilogue constants."""
    g = kc // 8
    tyb = mt * 4 * (16 // tz)
    hs16 = round_up((tyb + 1) * tz * g, 16)
    return ((9 * g + 3) // 4) * nt * 1024 + 6 * round_up(2 * hs16 * 16, 1024) + 3 * nt * 16 * 4 + 16


def gather_plans(kind, wshape, cls, q, es, kc, nreal, kreal, n=1) -> List["IgemmPlan"]:
    """The depth -9 candidates: the gathering marching kernel on the stride-(2,2,1) 3x3x1 bf16 launches that read the fine level and write the coarse one (strided convolution,
    data gradient of a transposed convolution): a workgroup owns a column of `tyb` coarse rows x `tz` slices and walks along x with a ring of FINE planes in LDS.
    tile = (x steps per workgroup, coarse rows per workgroup, tz)."""
    offs = [tuple(t[0]) for t in cls.taps]
    if es != 2 or tuple(cls.is_) != (2, 2, 1) or tuple(cls.os) != (1, 1, 1) or tuple(cls.oo) != (0, 0, 0) or offs != _TAPS_3x3x1 or kc != round_up(kreal, 8):
        return []
    nt = (nreal + 15) // 16
    out = []
    for (c, t, tz, mt) in sorted(GATHER_SHAPES):
        tyb = 64 * mt // tz
        if c != kc or t != nt or q[1] % tyb or q[2] % tz or gather_lds_bytes(kc, nt, tz, mt) > 158 * 1024:
            continue
        cols = n * (q[1] // tyb) * (q[2] // tz)
        for target in (512, 1024):
            nxs = max(1, min(q[0] // 8, -(-target // cols)))
            lx = -(-q[0] // nxs)
            pl = IgemmPlan(kind, cls, tuple(q), kc, nreal, kreal, (lx, tyb, tz), mt, nt, 1, kc, 1, (9 * (kc // 8) + 3) // 4, gather_lds_bytes(kc, nt, tz, mt), -9)
            if not any(o.tile == pl.tile and o.mtw == pl.mtw for o in out):
                out.append(pl)
    return out


def chain_plan(cin: int, compact: bool, q, n: int = 1, cmid: int = 16) -> Optional[dict]:
    """Plan of a chained marching launch (csrc/chain.hip: two stride-1 3x3x1 convolutions with the tensor between them in LDS; inference only) for the pairs it is
    instantiated and measured for (tools/bench_chain.py) — the compact one-channel network input (1 -> 16 -> 16) and a 32-channel input (32 -> 16 -> 1) on columns of 128 rows,
    16 -> 32 -> 32 with the residual convolution of the input on columns of 64 rows — or None.  A workgroup owns all rows of `tz` slices; x is cut into segments so that the
    launch has about one workgroup per CU."""
    X, Y, Z = q
    if cmid == 16 and Y == 128 and cin == (8 if compact else 32):
        tz, waves, mtw, lead = (4, 16, 2, 3) if compact else (2, 16, 1, 1)
    elif cmid == 32 and Y == 64 and cin == 16 and not compact:
        tz, waves, mtw, lead = 4, 8, 2, 1
    else:
        return None
    if X < 32 or Z % tz:
        return None
    cols = n * (Z // tz)
    nxs = max(1, min(X // 16, int(round(256.0 / cols))))
    return dict(tz=tz, waves=waves, mtw=mtw, lead=lead, lx=-(-X // nxs))


def chain_pack_plan(wshape, q, es, kc, n=1) -> Optional["IgemmPlan"]:
    """The packed-weight layout a chained launch reads for one of its two convolutions: that of a marching plan (depth -5) holding all output tiles."""
    cls = lattice_classes("conv_fwd", tuple(wshape[2:]), (1, 1, 1))[0]
    kreal, nreal = gemm_dims("conv_fwd", wshape)
    for pl in march_plans("conv_fwd", wshape, cls, q, es, kc, nreal, kreal, n):
        if pl.depth == -5 and pl.nt == (nreal + 15) // 16:
            pl.pack_map = pack_map(pl, wshape)
            return pl
    return None


# (input channels, main tiles, residual tiles, TZ, M-tiles per wave): marching launches with the ResidualUnit's 1x1x1 residual convolution riding along (csrc/mconv.hip MC_R)
MARCH_RES_SHAPES = {(16, 2, 2, 8, 4), (16, 2, 2, 8, 8), (16, 2, 2, 4, 4), (16, 2, 2, 4, 2), (64, 2, 2, 2, 1), (64, 2, 2, 2, 2)}


def march_res_plans(wshape, wshape_res, cls, q, es, kc, n=1) -> List["IgemmPlan"]:
    """Marching plans of a stride-1 3x3x1 forward convolution with `res_tiles` more output tiles of the 1x1x1 convolution `wshape_res` of the same input
    (vsseg_igemm_desc.res_tiles): the instantiated subset of `march_plans`, each with the residual convolution's packed weights in `pack_map_res`."""
    kreal, nreal = gemm_dims("conv_fwd", wshape)
    kr, nr_ch = gemm_dims("conv_fwd", wshape_res)
    if kr != kreal or tuple(wshape_res[2:]) != (1, 1, 1):
        return []
    nr = (nr_ch + 15) // 16
    out = []
    for pl in march_plans("conv_fwd", wshape, cls, q, es, kc, nreal, kreal, n):
        if (kc, pl.nt, nr, pl.tile[2], pl.mtw) in MARCH_RES_SHAPES:
            r = dataclasses.replace(pl, res_tiles=nr)
            r.pack_map = pack_map(r, wshape)
            r.pack_map_res = residual_tile_pack_map(kc, nr, wshape_res)
            out.append(r)
    return out


def residual_tile_pack_map(kc, nr, wshape_res) -> np.ndarray:
    """int32 gather map [K-steps of the centre tap][nr][64][8] -> flat index into the 1x1x1 weight (or -1): the K order is the 3x3x1 launch's (tap, 8-channel group),
    restricted to the K-steps [G, ceil(5G / 4)) that hold the centre tap's groups [4G, 5G) (csrc/mconv.hip KLO / KHI)."""
    g_ = kc // 8
    klo, khi = g_, (5 * g_ + 3) // 4
    kreal, nreal = gemm_dims("conv_fwd", wshape_res)
    ks, t, lane, j = np.meshgrid(np.arange(klo, khi), np.arange(nr), np.arange(64), np.arange(8), indexing="ij")
    p = ks * 4 + (lane >> 4)
    c = (p - 4 * g_) * 8 + j
    nn = t * 16 + (lane & 15)
    valid = (p >= 4 * g_) & (p < 5 * g_) & (c < kreal) & (nn < nreal)
    flat = weight_flat_index("conv_fwd", wshape_res, np.where(valid, c, 0), np.where(valid, nn, 0), 0)
    return np.where(valid, flat, -1).astype(np.int32).reshape(-1)


# (H channels, P channels as stored, TZ, M-tiles per wave) instantiated by csrc/mwgrad.hip; rows per workgroup = 64 * mt / tz
MARCH_WGRAD_SHAPES = {(16, 16, 4, 4), (16, 16, 4, 2), (16, 16, 8, 4), (16, 16, 8, 8), (32, 16, 4, 4), (32, 16, 4, 2), (32, 16, 2, 2), (32, 8, 4, 4), (32, 8, 4, 2),
                      (16, 32, 4, 4), (16, 32, 8, 4), (16, 32, 4, 2), (32, 32, 4, 4), (32, 32, 4, 2), (32, 32, 2, 2), (64, 32, 2, 2), (64, 32, 2, 1), (64, 32, 4, 2)}


def march_wgrad_tiles(ch, cp, q, n, scratch_elems=48 * 1024 * 1024) -> List[Tuple[int, int, int]]:
    """Candidate tiles (x steps per workgroup, rows, z slices) of the marching weight-gradient kernel for a stride-1 3x3x1 bf16 convolution with
    `ch` input channels (H = X) and a dY stored with `cp` channels: about 512 / 1024 / 2048 workgroups, each with its own partial-sum slab."""
    out = []
    per_blk = (ch // 16) * 9 * max(1, cp // 16) * 256
    for (h, p, tz, mt) in sorted(MARCH_WGRAD_SHAPES):
        tyb = 64 * mt // tz
        if h != ch or p != cp or q[1] % tyb or q[2] % tz:
            continue
        cols = n * (q[1] // tyb) * (q[2] // tz)
        for target in (512, 1024, 2048):
            nxs = max(1, min(q[0] // 8, -(-target // cols)))
            lx = -(-q[0] // nxs)
            grid = cols * -(-q[0] // lx)
            if grid * per_blk > scratch_elems or grid > 2048 or (lx, tyb, tz) in out:
                continue
            out.append((lx, tyb, tz))
    return out


# (channels of y / dA = conv outputs, channels of x = conv inputs, TZ, M-tiles per wave) instantiated by csrc/mbwd.hip (vsseg_conv_bwd_fused)
FUSED_BWD_SHAPES = {(16, 16, 8, 4), (16, 16, 4, 4), (16, 16, 4, 2), (16, 16, 8, 8), (32, 16, 4, 4), (32, 16, 4, 2), (32, 16, 8, 4), (32, 32, 4, 4), (32, 32, 4, 2), (32, 32, 2, 2),
                    (32, 64, 4, 2), (32, 64, 2, 2), (32, 64, 2, 1)}


def fused_bwd_lds_bytes(cy, cx, tz, mt, res=False) -> int:
    """csrc/mbwd.hip mb_lds_bytes: dy ring (+ two raw planes of the residual gradient), two x planes, the data gradient's packed weights."""
    rows = mt * 4 * (16 // tz) + 2
    return (MARCH_RING + (2 if res else 0)) * round_up(rows * tz * (cy // 8) * 16, 256) + 2 * round_up((rows - 2) * tz * (cx // 8) * 16, 256) + ((9 * (cy // 8) + 3) // 4) * (cx // 16) * 1024


def fused_bwd_tiles(cout, cin, q, n, scratch_elems=48 * 1024 * 1024, res=False) -> List[Tuple[int, int, int]]:
    """Candidate tiles (x steps per workgroup, rows, z slices) of the fused BatchNorm-backward + data gradient + weight gradient launch of a stride-1 3x3x1
    convolution block (csrc/mbwd.hip): about 512 / 1024 / 2048 workgroups, each with its own weight-gradient slab."""
    out = []
    per_blk = (cout // 16) * (10 if res else 9) * (cin // 16) * 256
    for (cy, cx, tz, mt) in sorted(FUSED_BWD_SHAPES):
        tyb = 64 * mt // tz
        if cy != cout or cx != cin or q[1] % tyb or q[2] % tz or fused_bwd_lds_bytes(cy, cx, tz, mt, res) > 160 * 1024:
            continue
        cols = n * (q[1] // tyb) * (q[2] // tz)
        for target in (512, 1024, 2048):
            nxs = max(1, min(q[0] // 8, -(-target // cols)))
            lx = -(-q[0] // nxs)
            grid = cols * -(-q[0] // lx)
            if grid * per_blk > scratch_elems or grid > 4096 or (lx, tyb, tz) in out:
                continue
            out.append((lx, tyb, tz))
    return out


def residual_dgrad_pack_plan(wshape, q) -> "IgemmPlan":
    """Packed-weight layout [ksteps][nt][64][8] of the data gradient of a 1x1x1 convolution (K = cout in one chunk -> N = cin in one workgroup): what
    vsseg_conv_bwd_fused reads as `wpack_res` (the residual convolution riding along)."""
    cls = lattice_classes("conv_dgrad", (1, 1, 1), (1, 1, 1))[0]
    kreal, nreal = gemm_dims("conv_dgrad", wshape)
    kc = round_up(kreal, 8)
    nt = (nreal + 15) // 16
    pl = IgemmPlan("conv_dgrad", cls, tuple(q), kc, nreal, kreal, (4, 4, 4), 1, nt, 1, kc, 1, (kc // 8 + 3) // 4, 0, 1)
    pl.pack_map = pack_map(pl, wshape)
    return pl


# ---- fused output-parity classes on the streaming kernel ("pixel shuffle"): depth -4 ----------------------------------------
def shuffle_plans(kind, wshape, kernel, stride, q, es, kc, nreal, kreal) -> Optional[List["IgemmPlan"]]:
    """The output-parity classes of a stride-(2,2,1) 3x3x1 transposed convolution / data gradient fused into streaming-kernel launches: a
    stride-1 convolution on the coarse lattice over the 2x2x1 neighbourhood (+0 / +1) whose 4 channel tiles are parity classes — channel tile
    t of class (px, py) is stored at fine voxel (2x + px, 2y + py, z); a (class, tap) pair without a kernel element has zero weights.
    16 output channels: ONE launch for all four classes (9 of the 16 pairs are real).  32 output channels: one launch per px with the classes
    (px, 0), (px, 1) (two channel tiles each).  The per-class launches each read the whole input; these read it once / twice."""
    if kind not in ("convT_fwd", "conv_dgrad") or tuple(kernel) != (3, 3, 1) or tuple(stride) != (2, 2, 1) or es != 2:
        return None
    if (nreal, kc) not in ((16, 16), (16, 32), (32, 32), (32, 48)) or kc != kreal or any(v % t for v, t in zip(q, (8, 8, 4))):  # instantiations of sconv.hip (taps 4)
        return None
    classes = lattice_classes(kind, kernel, stride)
    if [tuple(c.oo) for c in classes] != [(0, 0, 0), (0, 1, 0), (1, 0, 0), (1, 1, 0)]:
        return None
    taps = [((dx, dy, 0), (0, 0, 0)) for dx in (0, 1) for dy in (0, 1)]
    g = kc // 8
    tpc = nreal // 16  # 16-channel tiles per parity class
    kidx = np.full((4, 4), -1, np.int64)  # [class][tap] -> flat kernel index
    for ci, cl in enumerate(classes):
        for off, w in cl.taps:
            kidx[ci, off[0] * 2 + off[1]] = (w[0] * wshape[3] + w[1]) * wshape[4] + w[2]
    plans = []
    for px in range(tpc):  # tpc == 1: all four classes in one launch; tpc == 2: classes (px, 0), (px, 1)
        cls = LatticeClass((2, 2, 1), (px, 0, 0), (1, 1, 1), taps)
        pl = IgemmPlan(kind, cls, tuple(q), kc, nreal, kreal, (8, 8, 4), 4, 4, 1, kc, 1, g, stream_lds_bytes(kc, 4, 4), -4)
        ks, t, lane, j = np.meshgrid(np.arange(g), np.arange(4), np.arange(64), np.arange(8), indexing="ij")
        p = ks * 4 + (lane >> 4)
        tap, cg = p // g, p % g
        c, n = cg * 8 + j, (t % tpc) * 16 + (lane & 15)
        d = kidx[2 * px * (tpc - 1) + t // tpc, tap]
        valid = (d >= 0) & (c < kreal) & (n < nreal)
        flat = weight_flat_index(kind, wshape, np.where(valid, c, 0), np.where(valid, n, 0), np.where(valid, d, 0))
        pl.pack_map = np.where(valid, flat, -1).astype(np.int32).reshape(-1)
        plans.append(pl)
    return plans


def is_shuffle(pl) -> bool:
    """A launch that computes fused output-parity classes (pixel shuffle): the streaming kernel's depth -4 plans and their marching variants."""
    return pl.depth == -4 or (pl.depth in MARCH_DEPTHS and tuple(pl.cls.os) == (2, 2, 1))


def march_shuffle_all_plans(kind, wshape, kernel, stride, q, es, kc, nreal, kreal, n=1) -> List["IgemmPlan"]:
    """ALL four parity classes of a stride-(2,2,1) 3x3x1 transposed convolution with 32 output channels as ONE marching launch (csrc/mconv.hip, TPC = 2: 48 input
    channels -> 4 classes x 2 tiles; the level-2 -> level-1 transposed convolution) where shuffle_plans needs one streaming launch per px, each reading the whole input.
    Channel tile t is tile t % 2 of class t // 2; packed weights [6 K-steps][8 tiles][64][8] over the 2x2x1 tap neighbourhood."""
    if kind not in ("convT_fwd", "conv_dgrad") or tuple(kernel) != (3, 3, 1) or tuple(stride) != (2, 2, 1) or es != 2 or (kind, nreal, kc) not in (("convT_fwd", 32, 48), ("conv_dgrad", 32, 32)) or kc != kreal:
        return []  # (the level-2 -> level-1 transposed convolution; the data gradient of the level-1 -> level-2 strided convolution)
    classes = lattice_classes(kind, kernel, stride)
    if [tuple(c.oo) for c in classes] != [(0, 0, 0), (0, 1, 0), (1, 0, 0), (1, 1, 0)]:
        return []
    taps = [((dx, dy, 0), (0, 0, 0)) for dx in (0, 1) for dy in (0, 1)]
    g, tpc, nt = kc // 8, 2, 8
    ksteps = (4 * g + 3) // 4
    kidx = np.full((4, 4), -1, np.int64)  # [class][tap] -> flat kernel index
    for ci, cl in enumerate(classes):
        for off, w in cl.taps:
            kidx[ci, off[0] * 2 + off[1]] = (w[0] * wshape[3] + w[1]) * wshape[4] + w[2]
    ks, t, lane, j = np.meshgrid(np.arange(ksteps), np.arange(nt), np.arange(64), np.arange(8), indexing="ij")
    p = ks * 4 + (lane >> 4)
    tap, cg = np.minimum(p // g, 3), p % g
    c, nn = cg * 8 + j, (t % tpc) * 16 + (lane & 15)
    d = kidx[t // tpc, tap]
    valid = (p < 4 * g) & (d >= 0) & (c < kreal) & (nn < nreal)
    flat = weight_flat_index(kind, wshape, np.where(valid, c, 0), np.where(valid, nn, 0), np.where(valid, d, 0))
    pm = np.where(valid, flat, -1).astype(np.int32).reshape(-1)
    cls = LatticeClass((2, 2, 1), (0, 0, 0), (1, 1, 1), taps)
    out = []
    for (tz, mt) in ((4, 2), (8, 4), (2, 1), (4, 4)):  # the MC_P entries of csrc/mconv.hip
        tyb = 64 * mt // tz
        rows = mt * 4 * (16 // tz) + 2
        lds = ksteps * nt * 1024 + MARCH_RING * round_up(rows * tz * g * 16, 256) + 5 * nt * 16 * 4 + 16
        if q[1] % tyb or q[2] % tz or lds > 160 * 1024:
            continue
        cols = n * (q[1] // tyb) * (q[2] // tz)
        for target in (512, 1024):
            nxs = max(1, min(q[0] // 8, -(-target // cols)))
            lx = -(-q[0] // nxs)
            pl = IgemmPlan(kind, cls, tuple(q), kc, nreal, kreal, (lx, tyb, tz), mt, nt, 1, kc, 1, ksteps, lds, -5)
            pl.pack_map = pm
            if not any(o.tile == pl.tile and o.mtw == pl.mtw for o in out):
                out.append(pl)
    return out


def march_shuffle_plans(sp: "IgemmPlan", n=1) -> List["IgemmPlan"]:
    """Marching-kernel variants (csrc/mconv.hip PS) of a fused-parity-classes plan of shuffle_plans: 32 input channels -> four classes of 16 channels (the level-1 ->
    level-0 transposed convolution).  Same lattice class, taps and packed weights; tile = (coarse x steps per workgroup, coarse rows, z slices); every fine
    output row is written as tz consecutive 32-byte voxels instead of the streaming kernel's 8x8x4 tiles."""
    if sp.depth != -4 or (sp.kind, sp.kc) not in (("convT_fwd", 32), ("conv_dgrad", 16)) or sp.nc != 16 or sp.nt != 4 or tuple(sp.cls.oo) != (0, 0, 0):
        return []  # (the level-1 -> level-0 transposed convolution; the data gradient of the level-0 -> level-1 strided convolution)
    q, out = sp.q, []
    if sp.kc == 16:  # entries that exist for pixel-shuffle launches only (MC_P of csrc/mconv.hip)
        g, ksteps = 2, 2
        for (tz, mt) in ((8, 4), (4, 2), (4, 4), (8, 8)):
            tyb = 64 * mt // tz
            rows = mt * 4 * (16 // tz) + 2
            lds = ksteps * 4 * 1024 + MARCH_RING * round_up(rows * tz * g * 16, 256) + 5 * 4 * 16 * 4 + 16
            if q[1] % tyb or q[2] % tz:
                continue
            cols = n * (q[1] // tyb) * (q[2] // tz)
            for target in (512, 1024):
                nxs = max(1, min(q[0] // 8, -(-target // cols)))
                pl = dataclasses.replace(sp, tile=(-(-q[0] // nxs), tyb, tz), mtw=mt, lds=lds, depth=-5)
                if not any(o.tile == pl.tile and o.mtw == pl.mtw for o in out):
                    out.append(pl)
        return out
    for (c, t, tz, mt) in sorted(MARCH_SHAPES):
        tyb = 64 * mt // tz
        if (c, t) != (32, 4) or q[1] % tyb or q[2] % tz:
            continue
        cols = n * (q[1] // tyb) * (q[2] // tz)
        for target in (512, 1024):
            nxs = max(1, min(q[0] // 8, -(-target // cols)))
            lx = -(-q[0] // nxs)
            for depth in ((-5, -6) if (c, t, tz, mt) in MARCH_WREG_SHAPES else (-5,)):
                lds = march_lds_bytes(32, 4, tz, mt, depth == -6) - (0 if depth == -6 else 5 * 4 * 1024)  # 4 of the 9 K-steps of weights are staged
                pl = dataclasses.replace(sp, tile=(lx, tyb, tz), mtw=mt, lds=lds, depth=depth)
                if not any(o.tile == pl.tile and o.mtw == pl.mtw and o.depth == pl.depth for o in out):
                    out.append(pl)
    return out


# ---- compute-bound kernel (csrc/cconv.hip): depth -3 -----------------------------------------------------------------------
COMPUTE_TILE = (4, 8, 16)
_TAPS_3x3x3 = [(t // 9 - 1, (t // 3) % 3 - 1, t % 3 - 1) for t in range(27)]


def class_split_plans(kind, wshape, kernel, stride, q, es, kc, nreal, kreal, aux_es=4, in_split=0, limit=6) -> Optional[List["IgemmPlan"]]:
    """All output-parity classes of a transposed convolution / strided data gradient as ONE launch of the general kernel: workgroup
    row s computes class s (all output channels) with its own taps and K-step count and stores to its own output offset
    (vsseg_igemm_desc.class_split).  The union of the classes' input offsets is the halo every workgroup fetches ({0,1}^3 for the
    3x3x3 stride-2 transitions of the deep levels: 8 launches of a few workgroups each become one)."""
    classes = lattice_classes(kind, kernel, stride)
    nt = (nreal + 15) // 16
    if not 2 <= len(classes) <= 8 or nt > 6 or in_split_unsupported(in_split, kc):
        return None
    offs = sorted({off for c in classes for off, _ in c.taps})
    if len(offs) > 8 or max(len(c.taps) for c in classes) > 8:
        return None
    union = LatticeClass(classes[0].os, (0, 0, 0), classes[0].is_, [(off, (0, 0, 0)) for off in offs])
    nvox = q[0] * q[1] * q[2]
    cks = sorted({c for c in range(8, kc + 1, 8) if kc % c == 0 and (not in_split or c == kc or in_split % c == 0)}, reverse=True)
    out = []
    for mtw in (2, 4, 1):
        if (nt >= 5 and mtw >= 4) or (mtw > 1 and nvox < 64 * mtw):
            continue
        tile = choose_tile(q, union.taps, 64 * mtw)
        for ck in cks:
            pl = _mk_plan(kind, wshape, union, q, es, kc, nreal, kreal, tile, mtw, nt, len(classes), ck, aux_es)
            if pl is not None:
                pl.classes = classes
                pl.pack_map = pack_map(pl, wshape)
                out.append(pl)
                if len([p for p in out if p.mtw == mtw]) >= 2:
                    break
    return out[:limit] or None


def compute_split(nreal):
    """(nt, nsplit) of the compute-bound kernel for `nreal` output channels: 2 or 3 sixteen-channel tiles per workgroup, 1 or 2 workgroups
    (4 for the 128-channel concat gradient of level 3) per voxel tile (cc_check() in csrc/cconv.hip)."""
    return {32: (2, 1), 48: (3, 1), 64: (2, 2), 96: (3, 2), 128: (2, 4)}.get(nreal)


def compute_lds_bytes(nt):
    return 2 * (34 * 1024 + 14 * nt * 1024) + 5 * nt * 16 * 4 + 16


def compute_plan(kind, wshape, cls, q, es, kc, nreal, kreal, in_split=0) -> Optional["IgemmPlan"]:
    """The depth -3 candidate: the compile-time-geometry kernel for the MFMA-bound launches (stride-1 3x3x3 bf16, input channels a multiple
    of 16 processed in 16-channel chunks, 32 / 48 / 64 / 96 output channels, extents divisible by the 4x8x16 tile)."""
    offs = [tuple(t[0]) for t in cls.taps]
    if es != 2 or tuple(cls.is_) != (1, 1, 1) or tuple(cls.os) != (1, 1, 1) or tuple(cls.oo) != (0, 0, 0) or offs != _TAPS_3x3x3:
        return None
    if any(v % t for v, t in zip(q, COMPUTE_TILE)) or kc % 16 or (kc < 32 and kreal != 1) or kc != round_up(kreal, 16) or (in_split and in_split % 16):  # (one real channel in a 16-channel chunk: the data gradient of a C -> 1 convolution)
        return None
    sp = compute_split(nreal)
    if sp is None:
        return None
    nt, nsplit = sp
    return IgemmPlan(kind, cls, tuple(q), kc, nreal, kreal, COMPUTE_TILE, 8, nt, nsplit, 16, kc // 16, 14, compute_lds_bytes(nt), -3)


# ---- deep-level kernel (csrc/dconv.hip): depth -7 -------------------------------------------------------------------------
DEEP_MAX_VOXELS = 1 << 18  # lattice voxels x batch up to which the deep-level kernel is offered (level 3 at batch 4: 196 608)
DEEP_WAVES = 4


def deep_red_tiles(mt, nt):
    return mt * nt if mt * nt <= 24 else (mt * nt + 1) // 2


def deep_lds_bytes(tile, is_, taps, ck, ksteps, mt, nt, classes: int) -> int:
    """Mirror of dc_check() in csrc/dconv.hip (`classes` = number of parity classes of the launch, 0 for an ordinary one): K-group table (one per class) | epilogue constants | statistics rows | halo (voxel stride padded to an odd number of 16-byte
    units) | the four waves' accumulator slabs (their own space when every parity class re-reads the halo, else the halo's)."""
    halo = 1
    for a in range(3):
        offs = [t[0][a] for t in taps]
        halo *= (tile[a] - 1) * is_[a] + (max(offs) - min(offs) + 1)
    vs = ((ck // 8) | 1) * 16
    hb, rb = round_up(halo * vs, 1024), DEEP_WAVES * deep_red_tiles(mt, nt) * 1024  # (whole 1 KiB DMA rows)
    off = round_up(ksteps * 16 * max(1, classes), 16) + 3 * nt * 64 + DEEP_WAVES * 2 * nt * 64
    return off + (hb + rb if classes else max(hb, rb))


def _deep_ok(tile, mt, nt):
    return all(t & (t - 1) == 0 and t <= 128 for t in tile) and tile[0] * tile[1] * tile[2] == 16 * mt and mt * nt <= 40


def deep_plans(kind, wshape, cls, q, es, kc, nreal, kreal, n=1, in_split=0, limit=12) -> List["IgemmPlan"]:
    """The depth -7 candidates: the deep-level kernel (csrc/dconv.hip) on the small bf16 launches — workgroups of 16 * mt lattice voxels x nt channel tiles whose
    four waves split the K-steps, weight fragments straight from L2.  Any lattice class (stride-1, strided, one parity class of a transposed convolution)."""
    nvox = n * q[0] * q[1] * q[2]
    if es != 2 or nvox > DEEP_MAX_VOXELS or kc % 8 or (in_split and in_split % 8):
        return []
    nt_total = (nreal + 15) // 16
    cks = sorted({c for c in range(8, kc + 1, 8) if kc % c == 0}, reverse=True)
    out = []
    for mt in (8, 4, 2):
        if nvox < 16 * mt:
            continue
        tile = choose_tile(q, cls.taps, 16 * mt)
        tiles = n * int(np.prod([-(-q[a] // tile[a]) for a in range(3)]))
        got = 0
        for nsplit in range(1, nt_total + 1):
            nt = -(-nt_total // nsplit)
            if nt > 6 or not _deep_ok(tile, mt, nt) or (nsplit > 1 and -(-nt_total // (nsplit - 1)) == nt):
                continue
            for ck in cks:
                ksteps = (len(cls.taps) * (ck // 8) + 3) // 4
                lds = deep_lds_bytes(tile, cls.is_, cls.taps, ck, ksteps, mt, nt, 0)
                if lds > LDS_LIMIT:
                    continue
                pl = IgemmPlan(kind, cls, tuple(q), kc, nreal, kreal, tuple(tile), mt, nt, nsplit, ck, kc // ck, ksteps, lds, -7)
                pl.pack_map = pack_map(pl, wshape)
                out.append((tiles * nsplit, pl))
                got += 1
                break  # the largest chunk that fits: fewest passes over the K loop
            if got >= (4 if tiles < 96 else 2):  # a launch of a few dozen tiles (level 5, an eval window's level 4): also the finer output-channel splits — more workgroups, each streaming a smaller slice of the weights
                break
    # prefer launches that fill the GPU once (256 CUs), then the larger tiles
    out.sort(key=lambda r: (abs(np.log2(max(r[0], 1) / 256.0)) > 1.5, -r[1].mtw, r[1].nsplit))
    return [pl for _, pl in out[:limit]]


def deep_class_plans(kind, wshape, kernel, stride, q, es, kc, nreal, kreal, n=1, in_split=0) -> List["IgemmPlan"]:
    """All output-parity classes of a transposed convolution / strided data gradient in ONE launch of the deep-level kernel: a workgroup loads the halo of its
    coarse tile once and runs the classes one after the other (each with its own taps, packed weights [class][1][ksteps][nt][64][8] as class_split_plans)."""
    nvox = n * q[0] * q[1] * q[2]
    if kind not in ("convT_fwd", "conv_dgrad") or es != 2 or nvox > DEEP_MAX_VOXELS or kc % 8 or (in_split and in_split % 8):
        return []
    classes = lattice_classes(kind, kernel, stride)
    nt = (nreal + 15) // 16
    if not 2 <= len(classes) <= 8 or nt > 6:
        return []
    offs = sorted({off for c in classes for off, _ in c.taps})
    if len(offs) > VSSEG_MAX_TAPS or max(len(c.taps) for c in classes) > 8:
        return []
    union = LatticeClass(classes[0].os, (0, 0, 0), classes[0].is_, [(off, (0, 0, 0)) for off in offs])
    ksteps = (max(len(c.taps) for c in classes) * (kc // 8) + 3) // 4
    out = []
    for mt in (8, 4, 2):
        if nvox < 16 * mt or not _deep_ok(choose_tile(q, union.taps, 16 * mt), mt, nt):
            continue
        tile = choose_tile(q, union.taps, 16 * mt)
        lds = deep_lds_bytes(tile, union.is_, union.taps, kc, ksteps, mt, nt, len(classes))
        if lds > LDS_LIMIT:
            continue
        pl = IgemmPlan(kind, union, tuple(q), kc, nreal, kreal, tuple(tile), mt, nt, len(classes), kc, 1, ksteps, lds, -7)
        pl.classes = classes
        pl.pack_map = pack_map(pl, wshape)
        out.append(pl)
    return out


# ---- transition kernel (csrc/tconv.hip): depth -8 -----------------------------------------------------------------------
TRANSITION_TILE = (4, 8, 8)


def transition_lds_bytes(kc: int) -> int:
    """Mirror of tc_lds_bytes() in csrc/tconv.hip: K-group table of the eight classes | epilogue constants | statistics rows | halo (5x9x9 coarse voxels, voxel stride padded
    to an odd number of 16-byte units, whole 1 KiB DMA rows) | two weight buffers of the 8-tap class."""
    cg = kc // 8
    ksmax = (8 * cg + 3) // 4
    off_epi = 8 * ksmax * 4 * 4
    off_stat = off_epi + 3 * 3 * 16 * 4
    off_halo = round_up(off_stat + 4 * 2 * 3 * 16 * 4, 1024)
    hrows = (5 * 9 * 9 * (cg | 1) + 63) // 64
    return off_halo + hrows * 1024 + 2 * ksmax * 3 * 1024


def transition_plans(kind, wshape, kernel, stride, q, es, kc, nreal, kreal, in_split=0) -> List["IgemmPlan"]:
    """All eight output-parity classes of a 3x3x3 stride-(2,2,2) transposed convolution / strided data gradient with 48 or 64 input and 48 output channels (the transition
    between the 96x32x128 and 48x16x64 levels) as ONE launch of the transition kernel: 4x8x8 coarse tiles, the waves split the voxels, a stage per class (csrc/tconv.hip)."""
    if kind not in ("convT_fwd", "conv_dgrad") or es != 2 or tuple(kernel) != (3, 3, 3) or tuple(stride) != (2, 2, 2) or kc not in (48, 64) or in_split or nreal != 48:
        return []
    if any(q[a] % TRANSITION_TILE[a] for a in range(3)):
        return []
    classes = lattice_classes(kind, kernel, stride)
    if len(classes) != 8 or max(len(c.taps) for c in classes) > 8:
        return []
    offs = sorted({off for c in classes for off, _ in c.taps})
    for a in range(3):
        lo, hi = min(o[a] for o in offs), max(o[a] for o in offs)
        if hi - lo != 1 or lo not in (0, -1):
            return []
    union = LatticeClass(classes[0].os, (0, 0, 0), classes[0].is_, [(off, (0, 0, 0)) for off in offs])
    pl = IgemmPlan(kind, union, tuple(q), kc, nreal, kreal, TRANSITION_TILE, 16, 3, 8, kc, 1, (8 * (kc // 8) + 3) // 4, transition_lds_bytes(kc), -8)
    pl.classes = classes
    pl.pack_map = pack_map(pl, wshape)
    return [pl]


VSSEG_MAX_TAPS = 27


def _pow2_floor(v):
    p = 1
    while p * 2 <= v:
        p *= 2
    return p


def choose_tile(q, taps, voxels):
    """Lattice tile with `voxels` points: keep the contiguous z run >= 4 and spend the rest where the halo is."""
    ext = [max(t[0][a] for t in taps) - min(t[0][a] for t in taps) for a in range(3)]
    cap = [max(1, _pow2_floor(max(1, v))) if v >= 1 else 1 for v in q]
    cap = [c if c >= qq else c * 2 for c, qq in zip(cap, q)]  # allow one partial tile (next pow2 >= q)
    tile = [1, 1, 1]
    tile[2] = min(cap[2], 4 if ext[2] == 0 else 8, voxels)
    rem = voxels // tile[2]
    # distribute the remainder over x,y (then z) as evenly as possible
    order = [1, 0, 2]
    while rem > 1:
        grew = False
        for a in sorted(order, key=lambda a: tile[a] / (1 + ext[a])):
            if tile[a] * 2 <= cap[a] and rem > 1:
                tile[a] *= 2
                rem //= 2
                grew = True
                break
        if not grew:  # lattice smaller than the tile: pad the smallest axis (keeps the halo of strided convs compact)
            a = min(range(3), key=lambda a: (tile[a], -a))
            tile[a] *= 2
            rem //= 2
    return tuple(tile)


def _mk_plan(kind, wshape, cls, q, es, kc, nreal, kreal, tile, mtw, nt, nsplit, ck, aux_es, depth=1) -> Optional[IgemmPlan]:
    ntaps = len(cls.taps)
    ksteps = (ntaps * (ck // 8) + 3) // 4
    lds = igemm_lds_bytes(tile, cls.is_, cls.taps, ck, ksteps, nt, mtw, es, kc // ck, aux_es, depth)
    if lds > LDS_LIMIT - 1024 or igemm_halo_bytes(tile, cls.is_, cls.taps, ck, es) > HALO_MAX:
        return None
    return IgemmPlan(kind, cls, tuple(q), kc, nreal, kreal, tuple(tile), mtw, nt, nsplit, ck, kc // ck, ksteps, lds, depth)


def plan_igemm(kind, wshape, cls: LatticeClass, q, es, kc_pad=None, lds_budget=158 * 1024, mtw=None, aux_es=4, in_split=0) -> IgemmPlan:
    """The default (heuristic) plan.  in_split: the input is a two-part tensor split at that channel — a channel chunk must
    not straddle the split unless it is the only chunk (igemm_kernel.h chooses the part per chunk, or per DMA piece when
    there is one chunk).  `candidate_plans` lists the alternatives the engine's autotuner measures against this one."""
    kreal, nreal = gemm_dims(kind, wshape)
    kc = round_up(kreal, 8) if kc_pad is None else kc_pad
    nt_total = (nreal + 15) // 16
    nsplit0 = (nt_total + 5) // 6
    nvox = q[0] * q[1] * q[2]
    if mtw is None:
        mtw0 = 4 if nvox >= 2048 else (2 if nvox >= 512 else 1)
        mtws = [m for m in (4, 2, 1) if m <= mtw0]
    else:
        mtws = [mtw]
    cands = sorted({c for c in range(8, kc + 1, 8) if kc % c == 0 and (not in_split or c == kc or in_split % c == 0)}, reverse=True)
    best = None
    for budget in (lds_budget, LDS_LIMIT - 1024):  # soft budget first, then whatever fits
        for nsplit in sorted({nsplit0, min(nt_total, 2 * nsplit0), min(nt_total, 3 * nsplit0)}):  # fewer channel tiles per workgroup when the weights do not fit
            nt = (nt_total + nsplit - 1) // nsplit
            for mtw_ in mtws:  # a smaller voxel tile when even the smallest channel chunk does not fit (stride-2 3x3x3 halos in fp32)
                if (nt >= 5 and mtw_ >= 4) or (mtw_ == 8 and (nt < 3 or aux_es)):  # 512-voxel tiles: 3-4 channel tiles per workgroup, no auxiliary tile (it would not fit twice)
                    continue
                tile = choose_tile(q, cls.taps, 64 * mtw_)
                for ck in cands:
                    pl = _mk_plan(kind, wshape, cls, q, es, kc, nreal, kreal, tile, mtw_, nt, nsplit, ck, aux_es)
                    if pl is not None and pl.lds <= budget:
                        best = pl
                        break
                if best:
                    break
            if best:
                break
        if best:
            break
    if best is None:
        raise ValueError(f"no LDS-feasible plan for {kind} w={tuple(wshape)} q={q}")
    best.pack_map = pack_map(best, wshape)
    return best


def in_split_unsupported(in_split, kc) -> bool:
    return bool(in_split) and (in_split % 8 != 0 or in_split >= kc)


def candidate_plans(kind, wshape, cls: LatticeClass, q, es, kc_pad=None, aux_es=4, in_split=0, limit=10, n=1) -> List[IgemmPlan]:
    """Feasible alternatives to `plan_igemm`'s choice, default first (pack maps are filled for all of them).

    What differs between them is what the heuristic cannot see without measuring: the channel chunk (LDS footprint, hence
    how many workgroups share a CU, against the number of pipeline stages per tile), the voxel tile (MTW) and the
    output-channel split.  The measured optimum moved by up to 1.7x between neighbouring candidates (tools/sweep_ck.sh)."""
    default = plan_igemm(kind, wshape, cls, q, es, kc_pad=kc_pad, aux_es=aux_es, in_split=in_split)
    kreal, nreal = gemm_dims(kind, wshape)
    kc = default.kc
    nt_total = (nreal + 15) // 16
    out, seen = [default], {(default.tile, default.nt, default.nsplit, default.ck)}
    cks = sorted({c for c in range(8, kc + 1, 8) if kc % c == 0 and (not in_split or c == kc or in_split % c == 0)}, reverse=True)
    nvox = q[0] * q[1] * q[2]
    nsplits = {default.nsplit, min(nt_total, default.nsplit + 1), max(1, default.nsplit - 1)}
    if kc <= 16:  # tiny K (z-folded 1-channel inputs): re-reading the input per split is free, small NT keeps 3-4 workgroups per CU
        nsplits |= {nt_total, (nt_total + 1) // 2}
    for nsplit in sorted(nsplits):
        nt = (nt_total + nsplit - 1) // nsplit
        if nt > 6:
            continue
        for mtw in (8, 4, 2):
            if (nt >= 5 and mtw >= 4) or nvox < 512 * (mtw // 2) or (mtw == 8 and (nt < 3 or aux_es or nvox < 512 * 256)):  # 512-voxel tiles: MFMA-bound layers with >= one tile per CU
                continue
            tile = choose_tile(q, cls.taps, 64 * mtw)
            for ck in cks:
                key = (tuple(tile), nt, nsplit, ck)
                if key in seen:
                    continue
                pl = _mk_plan(kind, wshape, cls, q, es, kc, nreal, kreal, tile, mtw, nt, nsplit, ck, aux_es)
                if pl is None:
                    continue
                seen.add(key)
                out.append(pl)
    # keep the default, then prefer few chunks / the default split; cap the list
    rest = sorted(out[1:], key=lambda p: (p.mtw != 8, p.nsplit != default.nsplit, p.nchunks, -p.mtw))[: limit - 1]  # the 512-voxel variants are always measured
    # no-prefetch twins (depth -1: one LDS buffer, about half the footprint) where registers allow more than one resident
    # workgroup: 32->16 full-res 0.85 -> 0.78 ms, 64->32 half-res 0.86 -> 0.61 ms (tools/sweep_depth0.sh)
    twins = []
    for pl in [default] + rest:
        if pl.depth == -1 or pl.nt >= 3:  # nt >= 3 runs producer / consumer waves: always at least double-buffered
            continue
        if pl.nt <= 2 or pl.mtw <= 2 or pl.mtw == 8:  # register budget allows a second resident workgroup (512-voxel tiles: one buffer is all that fits)
            lds = igemm_lds_bytes(pl.tile, cls.is_, cls.taps, pl.ck, pl.ksteps, pl.nt, pl.mtw, es, pl.nchunks, aux_es, -1)
            twins.append(dataclasses.replace(pl, depth=-1, lds=lds))
    rest = rest + twins
    sp = stream_plan(kind, wshape, cls, q, es, kc, nreal, kreal)
    if sp is not None and not in_split_unsupported(in_split, kc):
        rest = rest + [sp]
    cp = compute_plan(kind, wshape, cls, q, es, kc, nreal, kreal, in_split)
    if cp is not None:
        rest = rest + [cp]
    if not in_split_unsupported(in_split, kc):
        rest = rest + march_plans(kind, wshape, cls, q, es, kc, nreal, kreal, n)
    rest = rest + deep_plans(kind, wshape, cls, q, es, kc, nreal, kreal, n, in_split)
    if not in_split:
        rest = rest + gather_plans(kind, wshape, cls, q, es, kc, nreal, kreal, n)
    for pl in rest:
        if pl.pack_map is None:
            pl.pack_map = pack_map(pl, wshape)
    return [default] + rest


def pack_map(plan: IgemmPlan, wshape) -> np.ndarray:
    """int32 gather map [nsplit][nchunks][ksteps][nt][64][8] -> flat weight index (or -1)."""
    kreal, nreal = gemm_dims(plan.kind, wshape)
    kdims = wshape[2:]
    cgs = plan.ck // 8
    ntaps = plan.ntaps
    S, CH, KS, NT = plan.nsplit, plan.nchunks, plan.ksteps, plan.nt
    split, ch, ks, t, lane, j = np.meshgrid(np.arange(S), np.arange(CH), np.arange(KS), np.arange(NT), np.arange(64), np.arange(8), indexing="ij")
    g = lane >> 4
    p = ks * 4 + g
    tap = p // cgs
    cg = p % cgs
    c = ch * plan.ck + cg * 8 + j
    if plan.classes is not None:  # class-split: row `split` is lattice class `split` (its own taps, every output channel)
        n = t * 16 + (lane & 15)
        cnt = np.array([len(c_.taps) for c_ in plan.classes])[split]
        widx = np.zeros((S, 8), np.int64)
        for s_, c_ in enumerate(plan.classes):
            for i, (_, w) in enumerate(c_.taps):
                widx[s_, i] = (w[0] * kdims[1] + w[1]) * kdims[2] + w[2]
        valid = (tap < cnt) & (c < kreal) & (n < nreal)
        d = widx[split, np.clip(tap, 0, 7)]
    else:
        n = (split * NT + t) * 16 + (lane & 15)
        valid = (p < ntaps * cgs) & (c < kreal) & (n < nreal)
        tapc = np.clip(tap, 0, ntaps - 1)
        widx = np.array([(w[0] * kdims[1] + w[1]) * kdims[2] + w[2] for _, w in plan.cls.taps], dtype=np.int64)
        d = widx[tapc]
    flat = weight_flat_index(plan.kind, wshape, np.where(valid, c, 0), np.where(valid, n, 0), d)
    return np.where(valid, flat, -1).astype(np.int32).reshape(-1)


# ---- z-folding: convolutions without taps along z and with a single real input or output channel ---------------------
# A [N,X,Y,Z,C] tensor is bit-identical to [N,X,Y,Z/8,8*C] (folded channel = (z % 8)*C + c).  A kernel with kz == 1 maps
# z-slice j of the input to z-slice j of the output, so the convolution equals one with 8*Cin inputs, 8*Cout outputs and
# block-diagonal weights on the folded tensors.  For Cin == 1 or Cout == 1 this replaces the 8-fold zero-extended channel
# group the MFMA path needs by 8 real z-neighbours: the launch reads / writes 1/8 of the bytes on that side and the MFMA
# count is unchanged (the zeros move from padded channels to off-diagonal blocks).
FOLD = 8


def foldable(kernel, stride, transposed, cin, cout, dims) -> bool:
    return (not transposed) and kernel[2] == 1 and tuple(stride) == (1, 1, 1) and dims[2] % FOLD == 0 and (cin == 1 or cout == 1) and max(cin, cout) * FOLD <= 128


def folded_wshape(wshape) -> tuple:
    return (wshape[0] * FOLD, wshape[1] * FOLD, *wshape[2:])


def fold_pack_map(vmap: np.ndarray, wshape) -> np.ndarray:
    """Pack map of a plan made for `folded_wshape(wshape)` -> flat indices into the REAL weight (or -1 off the diagonal)."""
    cout, cin = int(wshape[0]), int(wshape[1])
    T = int(np.prod(wshape[2:]))
    v = vmap.astype(np.int64)
    ok = v >= 0
    vv = np.where(ok, v, 0)
    t = vv % T
    iv = (vv // T) % (cin * FOLD)
    ov = vv // (T * cin * FOLD)
    jo, co = ov // cout, ov % cout
    ji, ci = iv // cin, iv % cin
    real = (co * cin + ci) * T + t
    return np.where(ok & (jo == ji), real, -1).astype(np.int32)


def folded_candidate_plans(kind, wshape, cls: LatticeClass, q, es, aux_es=4, heuristic_only=False) -> List[IgemmPlan]:
    """`candidate_plans` of the z-folded formulation; q is the REAL lattice extent, the plans' q / kc / nc are folded."""
    vw = folded_wshape(wshape)
    qv = (q[0], q[1], q[2] // FOLD)
    kreal, _ = gemm_dims(kind, vw)
    if heuristic_only:
        cands = [plan_igemm(kind, vw, cls, qv, es, kc_pad=round_up(kreal, 8), aux_es=aux_es)]
    else:
        cands = candidate_plans(kind, vw, cls, qv, es, kc_pad=round_up(kreal, 8), aux_es=aux_es)
    for pl in cands:
        pl.pack_map = fold_pack_map(pl.pack_map, wshape)
    return cands


def pack_map_centre(plan: IgemmPlan, wshape_1x1) -> np.ndarray:
    """Gather map (same layout as `pack_map`) that places a 1x1x1 convolution's weights on the zero-offset tap of `plan`
    and -1 elsewhere: adding it to the plan's own map merges `conv_k(x) + conv_1x1(x)` into one convolution."""
    kreal, nreal = gemm_dims(plan.kind, wshape_1x1)
    cgs = plan.ck // 8
    S, CH, KS, NT = plan.nsplit, plan.nchunks, plan.ksteps, plan.nt
    split, ch, ks, t, lane, j = np.meshgrid(np.arange(S), np.arange(CH), np.arange(KS), np.arange(NT), np.arange(64), np.arange(8), indexing="ij")
    p = ks * 4 + (lane >> 4)
    tap = p // cgs
    c = ch * plan.ck + (p % cgs) * 8 + j
    n = (split * NT + t) * 16 + (lane & 15)
    centre = [i for i, (off, _) in enumerate(plan.cls.taps) if tuple(off) == (0, 0, 0)]
    assert len(centre) == 1, "the merged residual needs exactly one zero-offset tap"
    valid = (tap == centre[0]) & (c < kreal) & (n < nreal)
    flat = weight_flat_index(plan.kind, wshape_1x1, np.where(valid, c, 0), np.where(valid, n, 0), 0)
    return np.where(valid, flat, -1).astype(np.int32).reshape(-1)


def out_dims(kind, in_dims, kernel, stride):
    pad = same_pad(kernel)
    if kind == "conv_fwd":
        return tuple((d + 2 * p - k) // s + 1 for d, p, k, s in zip(in_dims, pad, kernel, stride))
    if kind == "convT_fwd":  # output_padding = s + 2p - (k-1) - 1  ->  out = in * s   (ref:.../convolutions.py:117-123)
        return tuple(d * s for d, s in zip(in_dims, stride))
    raise ValueError(kind)


def simulate_igemm(plan: IgemmPlan, x: np.ndarray, wflat: np.ndarray, out_shape) -> np.ndarray:
    """Literal numpy restatement of igemm_kernel's indexing: x [N,X,Y,Z,kc] -> out [N,*out_shape,nc] (float64 accumulate)."""
    N, X, Y, Z, C = x.shape
    assert C == plan.kc
    wpack = np.where(plan.pack_map >= 0, wflat[np.clip(plan.pack_map, 0, None)], 0.0).reshape(plan.nsplit, plan.nchunks, plan.ksteps, plan.nt, 64, 8)
    cls = plan.cls
    q = plan.q
    cgs = plan.ck // 8
    qi = [np.arange(q[a]) for a in range(3)]
    if plan.classes is not None:  # class-split: each workgroup row is a launch of its own class with the row's slice of the packed weights
        out = np.zeros((N, *out_shape, plan.nc), np.float64)
        for s_, c_ in enumerate(plan.classes):
            ks_c = (len(c_.taps) * cgs + 3) // 4  # the K steps this row runs (the kernel's my_ks)
            sub = dataclasses.replace(plan, cls=c_, classes=None, nsplit=1, ksteps=ks_c, pack_map=plan.pack_map.reshape(plan.nsplit, plan.nchunks, plan.ksteps, -1)[s_, :, :ks_c].reshape(-1))
            o = simulate_igemm(sub, x, wflat, out_shape)
            sel = np.ix_(np.arange(N), *[np.arange(c_.oo[a], out_shape[a], c_.os[a]) for a in range(3)], np.arange(plan.nc))
            out[sel] = o[sel]
        return out
    acc = np.zeros((N, *q, plan.nsplit * plan.nt * 16), np.float64)
    for ch in range(plan.nchunks):
        for ks in range(plan.ksteps):
            for g in range(4):
                p = ks * 4 + g
                if p >= plan.ntaps * cgs:
                    continue
                tap, cg = divmod(p, cgs)
                off = cls.taps[tap][0]
                idx = [qi[a] * cls.is_[a] + off[a] for a in range(3)]
                ok = [(idx[a] >= 0) & (idx[a] < (X, Y, Z)[a]) for a in range(3)]
                xs = x[:, np.clip(idx[0], 0, X - 1)][:, :, np.clip(idx[1], 0, Y - 1)][:, :, :, np.clip(idx[2], 0, Z - 1)][..., ch * plan.ck + cg * 8 : ch * plan.ck + cg * 8 + 8].astype(np.float64)
                mask = ok[0][:, None, None] & ok[1][None, :, None] & ok[2][None, None, :]
                xs = xs * mask[None, ..., None]
                for split in range(plan.nsplit):
                    for t in range(plan.nt):
                        W = wpack[split, ch, ks, t, g * 16 : (g + 1) * 16, :].astype(np.float64)  # [n16][j]
                        n0 = (split * plan.nt + t) * 16
                        acc[..., n0 : n0 + 16] += np.einsum("nxyzj,cj->nxyzc", xs, W)
    out = np.zeros((N, *out_shape, plan.nc), np.float64)
    o = [qi[a] * cls.os[a] + cls.oo[a] for a in range(3)]
    sel = [o[a] < out_shape[a] for a in range(3)]
    out[np.ix_(np.arange(N), o[0][sel[0]], o[1][sel[1]], o[2][sel[2]], np.arange(plan.nc))] = acc[np.ix_(np.arange(N), qi[0][sel[0]], qi[1][sel[1]], qi[2][sel[2]], np.arange(plan.nc))]
    return out


# ---------------------------------------------------------------------------------------------------------------
# weight gradient
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class WgradPlan:
    transposed: bool
    q: Tuple[int, int, int]
    hs: Tuple[int, int, int]
    taps: List[Tuple[Tuple[int, int, int], int]]  # (halo offset, flat kernel index)
    tile: Tuple[int, int, int]
    ntp: int
    cp_valid: int
    ch_valid: int
    stride_p: int
    stride_h: int
    stride_tap: int
    lds: int
    blocks: int


def plan_wgrad(transposed: bool, wshape, kernel, stride, lattice_dims, es, cp_valid=None, ch_valid=None) -> WgradPlan:
    """Conv3d: P = dY (co), H = X (ci), lattice = dY dims.  ConvTranspose3d: P = X (ci), H = dY (co), lattice = X dims."""
    pad = same_pad(kernel)
    K = int(np.prod(kernel))
    taps = [((dx - pad[0], dy - pad[1], dz - pad[2]), (dx * kernel[1] + dy) * kernel[2] + dz) for dx in range(kernel[0]) for dy in range(kernel[1]) for dz in range(kernel[2])]
    if not transposed:  # W[co][ci][K]
        cp, chn = wshape[0], wshape[1]
        stride_p, stride_h = wshape[1] * K, K
    else:  # W[ci][co][K]
        cp, chn = wshape[0], wshape[1]
        stride_p, stride_h = wshape[1] * K, K
    cp_valid = cp if cp_valid is None else cp_valid
    ch_valid = chn if ch_valid is None else ch_valid
    ntp = (cp_valid + 15) // 16
    assert ntp <= 6, "P channels > 96 are not supported by the wgrad kernel"
    q = tuple(lattice_dims)
    nvox = q[0] * q[1] * q[2]
    pseudo = [((t[0]), (0, 0, 0)) for t in taps]
    for tv in (256, 128, 64, 32):
        if tv > 32 and nvox < tv * 2:
            continue
        tile = choose_tile(q, pseudo, tv)
        halo = 1
        for a in range(3):
            offs = [t[0][a] for t in taps]
            halo *= (tile[a] - 1) * stride[a] + (max(offs) - min(offs) + 1)
        p_bytes, h_bytes = tv * ntp * 16 * es, halo * 16 * es
        tables = ((p_bytes // 16 + 255) // 256 + (h_bytes // 16 + 255) // 256) * 1024  # boundary-path coordinate tables
        lds = round_up(tv * 4, 16) + 2 * p_bytes + 2 * h_bytes + tables  # double-buffered tiles (LDS-DMA pipeline), mirrors vsseg_wgrad()
        if lds <= LDS_LIMIT - 1024 and p_bytes <= 12 * 256 * 16 and h_bytes <= 8 * 256 * 16:
            break
    else:
        raise ValueError("no LDS-feasible wgrad tile")
    hchunks = (ch_valid + 15) // 16
    blocks = max(1, min(1024 // hchunks, 512))
    return WgradPlan(transposed, q, tuple(stride), taps, tile, ntp, cp_valid, ch_valid, stride_p, stride_h, 1, lds, blocks)
