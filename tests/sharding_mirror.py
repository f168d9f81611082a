"""TEST INFRASTRUCTURE (not part of the product package): Python mirror of the pair-tile ownership rule of libreo_hip.so
(launch_k1 in csrc/kernels.hip) and of the linear algebra of its tallies, so that the
multi-GPU protocol can be exercised without a GPU.

G is sharded by work unit: the upper triangle of the gene x gene pair matrix is
cut into tiles of TILE_I rows x CHUNK_J columns; a unit = UNIT_H consecutive
row tiles x Q column chunks of one panel; units are numbered panel-major and
unit u belongs to shard u % world.  Every shard builds the class-table bits of
its own tiles (forward and mirror) into a zeroed table; the shards' bits are
disjoint, so ONE exchange gives every shard the whole table -- an in-place sum
of the table, or (what the library's RCCL path does: x_pack / x_expand_* in
kernels.hip, mirrored by pack_units / expand_units below) an all-gather of the
forward rectangles of every shard's own units, the mirror words being derived
on arrival -- and tallies / iteration passes then run unsharded
(src/RankCompV3.jl:403).
"""
from __future__ import annotations

import numpy as np

TILE_I = 32     # kTileI
TILE_J = 256    # kTileJ
RJ = 4          # kRJ (genes per lane in the wave form of the pair kernel, and in the tie-free workgroup form)
RJ_TIES = 2     # kRJTies (genes per lane in the tie-rich workgroup form)
RJ_WIDE = 2     # kRJWide / kRJWideTies: the wide forms (more than 65 535 samples)
RJ_WIDE_TIES = 1
UNIT_H = 32     # kUnitH
GENE_PAD = 1024 # kGenePad
SLOT_PAD = 32   # sample slots per block of the bit planes


FORM = "wave"   # which pair kernel launch_k1 selects: "wave" (two groups; more with shared per-group counts when S <= 65535 -- the
                # default), "wg" (the workgroup form: per-comparison recounts of more than two groups, or REO_K1_WAVE=0), "wide"
                # (the workgroup form for S > 65535: more than two groups, or REO_K1_WAVE=0)


def genes_per_lane(has_ties: bool, form: str | None = None) -> int:
    form = form or FORM
    if form == "wave":
        return RJ
    if form == "wide":
        return RJ_WIDE_TIES if has_ties else RJ_WIDE
    return RJ_TIES if has_ties else RJ


def geometry(G: int, sample_slots: int, has_ties: bool, form: str | None = None):
    """(Gp, CJ, Q) exactly as launch_k1 derives them."""
    Gp = (G + GENE_PAD - 1) // GENE_PAD * GENE_PAD
    CJ = TILE_J * genes_per_lane(has_ties, form)
    chunk_bytes = CJ * sample_slots * 2
    Q = 4 if chunk_bytes * 4 <= (2 << 20) else (2 if chunk_bytes * 2 <= (2 << 20) else 1)
    return Gp, CJ, Q


def sample_slots(group_sizes) -> int:
    """Every group is padded to whole blocks of 32 sample slots (transform.hip)."""
    return int(sum((int(n) + SLOT_PAD - 1) // SLOT_PAD * SLOT_PAD for n in group_sizes))


def tile_owner(G: int, slots: int, has_ties: bool, world: int, form: str | None = None) -> np.ndarray:
    """owner[it, jc] = shard that computes pair tile (rows 32*it.., columns CJ*jc..), or -1 if the
    tile lies strictly below the diagonal (it is the mirror of another tile)."""
    Gp, CJ, Q = geometry(G, slots, has_ties, form)
    NJ, NIT = (Gp + CJ - 1) // CJ, Gp // TILE_I
    NP = (NJ + Q - 1) // Q
    owner = np.full((NIT, NJ), -1, dtype=np.int32)
    gu = 0
    for p in range(NP):
        ni = min(NIT, (CJ // TILE_I) * Q * (p + 1))
        r = 0
        while r * UNIT_H < ni:
            for t in range(r * UNIT_H, min(ni, (r + 1) * UNIT_H)):
                for jc in range(p * Q, min(NJ, (p + 1) * Q)):
                    if (jc * CJ + CJ - 1) // 64 < (t * TILE_I) // 64:
                        continue
                    owner[t, jc] = gu % world
            gu += 1
            r += 1
    return owner


def owned_pair_mask(G: int, slots: int, has_ties: bool, rank: int, world: int, form: str | None = None) -> np.ndarray:
    """mask[i, j] (i != j) = True iff the unordered pair {i, j} is computed by shard `rank`
    (the shard then holds both the (i,j) bits and the mirrored (j,i) bits)."""
    Gp, CJ, Q = geometry(G, slots, has_ties, form)
    owner = tile_owner(G, slots, has_ties, world, form)
    i = np.arange(G)
    lo, hi = np.minimum(i[:, None], i[None, :]), np.maximum(i[:, None], i[None, :])
    m = owner[lo // TILE_I, hi // CJ] == rank
    np.fill_diagonal(m, False)
    return m


def raw_counters(code: np.ndarray, ref: np.ndarray, mask: np.ndarray | None = None) -> np.ndarray:
    """The 8 counters k2_tally writes per gene: marginals cL cH tL tH and cells LL LH HL HH over the
    reference genes, restricted to the pairs in `mask` (a shard's share)."""
    code = np.asarray(code)
    G = code.shape[0]
    sel = np.asarray(ref, dtype=bool)[None, :] & (code < 9)
    if mask is not None:
        sel = sel & mask
    ic, it = code // 3, code % 3
    out = np.zeros((G, 8), dtype=np.int32)
    cL, cH, tL, tH = (ic == 0) & sel, (ic == 2) & sel, (it == 0) & sel, (it == 2) & sel
    for k, m in enumerate((cL, cH, tL, tH, cL & tL, cL & tH, cH & tL, cH & tH)):
        out[:, k] = m.sum(axis=1)
    return out


def derive_tallies(raw: np.ndarray, ref: np.ndarray) -> np.ndarray:
    """k3_derive: the 9 tallies n11..n33 from the (summed) raw counters."""
    raw = np.asarray(raw, dtype=np.int64)
    ref = np.asarray(ref, dtype=bool)
    total = int(ref.sum()) - ref.astype(np.int64)
    cLt, cHt, tLt, tHt, LL, LH, HL, HH = raw.T
    c = np.zeros((raw.shape[0], 9), dtype=np.int64)
    c[:, 0], c[:, 2], c[:, 6], c[:, 8] = LL, LH, HL, HH
    c[:, 1] = cLt - LL - LH
    c[:, 7] = cHt - HL - HH
    c[:, 3] = tLt - LL - HL
    c[:, 5] = tHt - LH - HH
    c[:, 4] = total - (cLt + cHt + c[:, 3] + c[:, 5])
    return c.astype(np.int32)


def class_planes(code: np.ndarray, mask: np.ndarray | None = None) -> np.ndarray:
    """The four bit planes cL cH tL tH of the class table ([G, 4, G], one int32 per bit, unpacked), restricted to the
    pairs in `mask` (a shard's share; everything else stays zero, like the shard's zeroed table)."""
    code = np.asarray(code)
    sel = code < 9
    if mask is not None:
        sel = sel & mask
    ic, it = code // 3, code % 3
    return np.stack([(ic == 0) & sel, (ic == 2) & sel, (it == 0) & sel, (it == 2) & sel], axis=1).astype(np.int32)


def codes_from_planes(planes: np.ndarray) -> np.ndarray:
    """Inverse of class_planes on a complete table: class codes 0..8, 255 on the diagonal (k_decode)."""
    cl, ch, tl, th = (planes[:, k, :] for k in range(4))
    ic = np.where(cl > 0, 0, np.where(ch > 0, 2, 1))
    it = np.where(tl > 0, 0, np.where(th > 0, 2, 1))
    code = (3 * ic + it).astype(np.uint8)
    np.fill_diagonal(code, 255)
    return code


def unit_list(G: int, slots: int, has_ties: bool):
    """Every work unit of the build in launch_k1's order: (panel p, i-range r); unit u belongs to shard u % world.  A
    unit's forward words are the rectangle rows [1024 r, 1024 r + 1024) x columns [W p, W p + W), W = Q * CJ."""
    Gp, CJ, Q = geometry(G, slots, has_ties)
    NJ, NIT = (Gp + CJ - 1) // CJ, Gp // TILE_I
    NP = (NJ + Q - 1) // Q
    units = []
    for p in range(NP):
        ni = min(NIT, (CJ // TILE_I) * Q * (p + 1))
        units += [(p, r) for r in range((ni + UNIT_H - 1) // UNIT_H)]
    return units, Q * CJ


def wave_plan(G: int, slots: int, has_ties: bool, world: int, waves: int = 4):
    """The pipelined exchange of launch_k1: a shard's slots (unit = shard + slot * world) are counted and exchanged in
    `nwaves` waves of `mw` slots -- the same numbers on every shard.  Returns (nwaves, mw, slots per shard)."""
    units, _ = unit_list(G, slots, has_ties)
    maxu = max(1, (len(units) + world - 1) // world)
    nwaves = min(waves, 8, maxu) if world > 1 else 1
    mw = (maxu + nwaves - 1) // nwaves
    return nwaves, mw, maxu


def wave_of_unit(G: int, slots: int, has_ties: bool, world: int, waves: int = 4) -> np.ndarray:
    """wave in which each unit of the build is counted (by its owner)."""
    units, _ = unit_list(G, slots, has_ties)
    nwaves, mw, _ = wave_plan(G, slots, has_ties, world, waves)
    return np.array([min((u // world) // mw, nwaves - 1) for u in range(len(units))], dtype=np.int64)


def pack_units(planes: np.ndarray, G: int, slots: int, has_ties: bool, rank: int, world: int, m0: int = 0, mcnt: int | None = None) -> np.ndarray:
    """x_pack: the forward rectangles of this shard's units (all of them, or the slots m0 .. m0 + mcnt - 1 of one wave) as they
    stand in its (partial) table `planes` [G, 4, G], one after the other, zero-padded to whole rectangles and to the slot count."""
    units, W = unit_list(G, slots, has_ties)
    H = UNIT_H * TILE_I
    maxu = max(1, (len(units) + world - 1) // world) if mcnt is None else mcnt
    out = np.zeros((maxu, H, 4, W), dtype=planes.dtype)
    for m in range(maxu):
        gu = rank + (m0 + m) * world
        if gu >= len(units):
            continue
        p, r = units[gu]
        blk = planes[r * H:(r + 1) * H, :, p * W:(p + 1) * W]
        out[m, :blk.shape[0], :, :blk.shape[2]] = blk
    return out


def expand_units(planes: np.ndarray, recv: np.ndarray, G: int, slots: int, has_ties: bool, rank: int, world: int, m0: int = 0) -> None:
    """x_expand_fwd + x_expand_mirror: OR the other shards' rectangles (recv[s] = shard s's pack of the slots from m0 on) into
    `planes`, then their transposes with the low / high planes swapped (the pair seen from the other gene, src/RankCompV3.jl:386)."""
    units, W = unit_list(G, slots, has_ties)
    H = UNIT_H * TILE_I
    swap = [1, 0, 3, 2]
    todo = [(s, m, units[s + (m0 + m) * world]) for s in range(world) if s != rank for m in range(recv.shape[1]) if s + (m0 + m) * world < len(units)]
    for s, m, (p, r) in todo:
        rows, cols = slice(r * H, min((r + 1) * H, G)), slice(p * W, min((p + 1) * W, G))
        planes[rows, :, cols] |= recv[s, m, :rows.stop - rows.start, :, :cols.stop - cols.start]
    for s, m, (p, r) in todo:
        rows, cols = slice(r * H, min((r + 1) * H, G)), slice(p * W, min((p + 1) * W, G))
        blk = recv[s, m, :rows.stop - rows.start, :, :cols.stop - cols.start]
        for pl in range(4):
            planes[cols, swap[pl], rows] |= blk[:, pl, :].T
