"""Generate tests/golden/*.npz from the reference's Arrow IPC fixtures (run in the build container,
where /root/reference exists; the GPU box only sees the committed .npz files).

    python tests/golden/make_golden.py

Each .npz holds: the raw WKB column (values + offsets), the GeoArrow buffers decoded by an INDEPENDENT
pure-Python WKB reader (struct-based, below — not the library's decoder), known-answer columns that
ship with the fixture (nybb Shape_Area / Shape_Leng), and the CPU oracle's outputs as regression pins.
Fixtures: SURVEY.md §2 row 12.
"""
import os
import struct
import sys

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference"
FILES = {
    "cities": f"{REF}/data/cities.arrow",
    "naturalearth_cities": f"{REF}/py-geopolars/python/geopolars/datasets/naturalearth_cities.arrow",
    "naturalearth_lowres": f"{REF}/py-geopolars/python/geopolars/datasets/naturalearth_lowres.arrow",
    "nybb": f"{REF}/py-geopolars/python/geopolars/datasets/nybb.arrow",
}


def read_table(path):
    with open(path, "rb") as f:
        try:
            return ipc.open_file(f).read_all()
        except pa.ArrowInvalid:
            f.seek(0)
            return ipc.open_stream(f).read_all()


def parse_wkb(buf):
    """-> (type, parts) with parts = list of polygons, polygon = list of rings, ring = list of (x, y)."""
    pos = 0

    def rd(fmt):
        nonlocal pos
        v = struct.unpack_from(fmt, buf, pos)
        pos += struct.calcsize(fmt)
        return v

    def geom():
        (bo,) = rd("B")
        e = "<" if bo == 1 else ">"
        (t,) = rd(e + "I")
        assert t in (1, 2, 3, 4, 5, 6), t
        if t == 1:
            return t, [rd(e + "2d")]
        if t == 2:
            (n,) = rd(e + "I")
            return t, [rd(e + "2d") for _ in range(n)]
        if t == 3:
            (nr,) = rd(e + "I")
            rings = []
            for _ in range(nr):
                (n,) = rd(e + "I")
                rings.append([rd(e + "2d") for _ in range(n)])
            return t, rings
        (k,) = rd(e + "I")
        return t, [geom()[1] for _ in range(k)]

    return geom()


def decode_column(col):
    col = col.combine_chunks() if isinstance(col, pa.ChunkedArray) else col
    rows = [parse_wkb(v.as_py()) for v in col]
    types = {t for t, _ in rows}
    if types <= {1}:
        xy = np.array([g[0] for _, g in rows], dtype=np.float64)
        return dict(geom_type=0, xy=xy)
    assert types <= {3, 6}, types
    multi = 6 in types
    xy, ring_off, part_off, geom_off = [], [0], [0], [0]
    for t, g in rows:
        polys = g if t == 6 else [g]
        for rings in polys:
            for ring in rings:
                xy.extend(ring)
                ring_off.append(len(xy))
            part_off.append(len(ring_off) - 1)
        geom_off.append(len(part_off) - 1 if multi else len(ring_off) - 1)
    out = dict(geom_type=6 if multi else 3, xy=np.array(xy, dtype=np.float64), ring_offsets=np.array(ring_off, np.int32), geom_offsets=np.array(geom_off, np.int32))
    if multi:
        out["part_offsets"] = np.array(part_off, np.int32)
    return out


def main():
    from geopolars_amd.geoarrow import GeoArrowArray
    from oracle import pyoracle as O

    for name, path in FILES.items():
        tbl = read_table(path)
        col = tbl.column("geometry").combine_chunks()
        if pa.types.is_large_binary(col.type):
            col = col.cast(pa.binary())
        bufs = col.buffers()
        offsets = np.frombuffer(bufs[1], dtype=np.int32)[col.offset : col.offset + len(col) + 1].copy()
        values = np.frombuffer(bufs[2], dtype=np.uint8).copy()
        dec = decode_column(col)
        arr = GeoArrowArray(dec["geom_type"], dec["xy"], dec.get("geom_offsets"), dec.get("part_offsets"), dec.get("ring_offsets"))
        out = dict(wkb_values=values, wkb_offsets=offsets, **dec)
        out["oracle_bounds"] = O.bounds(arr)
        out["oracle_area"] = O.area(arr)
        c, v = O.centroid(arr)
        out["oracle_centroid"] = c
        out["oracle_length"] = O.euclidean_length(arr)
        for extra in ("Shape_Area", "Shape_Leng"):
            if extra in tbl.column_names:
                out[extra] = np.asarray(tbl.column(extra).to_pylist(), dtype=np.float64)
        if "name" in tbl.column_names:
            out["first_name"] = np.array(tbl.column("name")[0].as_py())
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(name, len(col), arr.n_coords, "coords", os.path.getsize(os.path.join(HERE, f"{name}.npz")), "bytes")


if __name__ == "__main__":
    main()
