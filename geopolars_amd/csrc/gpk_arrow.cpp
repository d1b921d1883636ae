// gpk_geoarray_from_arrow: the reference's one FFI seam — the Arrow C Data Interface (py-geopolars/src/ffi.rs:12-32 imports a pyarrow
// array through `_export_to_c`; :56,73,93 rechunk first, so there is ONE array) — as an entry point of this library.  Host code only:
// it walks the exported struct tree, finds the GeoArrow buffers in place (a sliced array's offsets are rebased, 64-bit offsets are
// narrowed after a range check, a validity bitmap with a bit offset is repacked) and hands them to gpk_geoarray_from_wkb (a WKB
// column, util.rs:27-37) or gpk_geoarray_upload (native GeoArrow nestings, interleaved or Struct<x, y> coordinates:
// py-geopolars/python/geopolars/internals/geoseries.py:86-113).  The array is BORROWED for the call and never released here.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gpk_common.h"

namespace {
using gpk::fail;

// "ARROW:extension:name" of a schema's metadata (int32 n, then n x (int32 len, key, int32 len, value), native endian), or ""
std::string extension_name(const ArrowSchema* s) {
    if (!s || !s->metadata) return std::string();
    const char* p = s->metadata;
    int32_t n = 0;
    memcpy(&n, p, 4);
    p += 4;
    // (the block carries no length of its own: a negative or absurd count / length is a producer's bug, not a reason to walk off the block)
    constexpr int32_t SANE = 1 << 24;
    if (n < 0 || n > SANE) return std::string();
    for (int32_t i = 0; i < n; ++i) {
        int32_t kl = 0, vl = 0;
        memcpy(&kl, p, 4);
        if (kl < 0 || kl > SANE) return std::string();
        const char* k = p + 4;
        memcpy(&vl, k + kl, 4);
        if (vl < 0 || vl > SANE) return std::string();
        const char* v = k + kl + 4;
        if (kl == 20 && !memcmp(k, "ARROW:extension:name", 20)) return std::string(v, (size_t)vl);
        p = v + vl;
    }
    return std::string();
}
bool fmt(const ArrowSchema* s, const char* f) { return s && s->format && !strcmp(s->format, f); }

// entries [lo, hi] of an offsets buffer (32- or 64-bit), rebased to start at 0, as i32
int32_t rebased_offsets(const void* buf, bool wide, int64_t lo, int64_t hi, std::vector<int32_t>& out, int64_t* first, int64_t* last, const char* what) {
    const int64_t n = hi - lo;
    if (n <= 0) {  // an empty array: producers may export it with a NULL or 0-byte offsets buffer — nothing is read
        out.assign(1, 0);
        *first = *last = 0;
        return GPK_OK;
    }
    if (!buf) return fail(GPK_ERR_INVALID_OFFSETS, "from_arrow: %s offsets buffer is NULL", what);
    out.resize((size_t)n + 1);
    int64_t f = 0, prev = 0;
    for (int64_t i = 0; i <= n; ++i) {
        const int64_t v = wide ? ((const int64_t*)buf)[lo + i] : (int64_t)((const int32_t*)buf)[lo + i];
        if (i == 0) f = prev = v;
        if (v < prev) return fail(GPK_ERR_INVALID_OFFSETS, "from_arrow: %s offsets decrease at entry %lld", what, (long long)i);
        if (v - f > (int64_t)INT32_MAX) return fail(GPK_ERR_INVALID_OFFSETS, "from_arrow: %s offsets beyond i32 (split the chunk)", what);
        out[(size_t)i] = (int32_t)(v - f);
        prev = v;
    }
    *first = f;
    *last = prev;
    return GPK_OK;
}
// n validity bits starting at bit `bit0` of `bits`, repacked to start at bit 0 (NULL in: NULL out)
const uint8_t* repacked_validity(const uint8_t* bits, int64_t bit0, int64_t n, std::vector<uint8_t>& store) {
    if (!bits) return nullptr;
    if (bit0 == 0) return bits;
    store.assign((size_t)((n + 7) / 8), 0);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t b = bit0 + i;
        if ((bits[b >> 3] >> (b & 7)) & 1) store[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
    }
    return store.data();
}
}  // namespace

extern "C" int32_t gpk_geoarray_from_arrow(const struct ArrowArray* array, const struct ArrowSchema* schema, int32_t geom_type_hint, void* stream,
                                           gpk_geoarray** out, int32_t* out_geom_type) {
    if (!array || !schema || !out) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    if (!schema->format) return fail(GPK_ERR_INVALID_ARGUMENT, "from_arrow: schema without a format string");
    if (array->length < 0 || array->offset < 0) return fail(GPK_ERR_INVALID_ARGUMENT, "from_arrow: negative length / offset");
    if (schema->dictionary || array->dictionary) return fail(GPK_ERR_MISMATCHED_GEOMETRY, "from_arrow: dictionary-encoded columns are not geometry columns");
    const std::string ext = extension_name(schema);
    const int64_t n = array->length, off = array->offset;
    std::vector<uint8_t> vstore;
    const uint8_t* validity = nullptr;
    if (array->n_buffers >= 1 && array->buffers && array->null_count != 0) validity = repacked_validity((const uint8_t*)array->buffers[0], off, n, vstore);

    // ---- a WKB column: Binary ("z", i32 offsets — from_geom_vec, util.rs:11-24) or LargeBinary ("Z" — what polars itself holds)
    if (fmt(schema, "z") || fmt(schema, "Z")) {
        if (array->n_buffers < 3 || !array->buffers) return fail(GPK_ERR_INVALID_ARGUMENT, "from_arrow: a binary array has three buffers");
        std::vector<int32_t> offs;
        int64_t first = 0, last = 0;
        GPK_TRY(rebased_offsets(array->buffers[1], fmt(schema, "Z"), off, off + n, offs, &first, &last, "binary"));
        const uint8_t* values = (const uint8_t*)array->buffers[2];
        if (!values && last > first) return fail(GPK_ERR_INVALID_ARGUMENT, "from_arrow: binary values buffer is NULL");
        return gpk_geoarray_from_wkb(values ? values + first : nullptr, offs.data(), n, validity, GPK_MEM_HOST, stream, out, out_geom_type);
    }

    // ---- native GeoArrow: List* around a coordinate array
    std::vector<int32_t> level[3];
    int depth = 0;
    const ArrowArray* a = array;
    const ArrowSchema* s = schema;
    int64_t lo = off, hi = off + n;  // PHYSICAL slots of `a` the column covers
    while (fmt(s, "+l") || fmt(s, "+L")) {
        if (depth == 3) return fail(GPK_ERR_MISMATCHED_GEOMETRY, "from_arrow: more than three list levels");
        if (a->n_buffers < 2 || !a->buffers || a->n_children != 1 || s->n_children != 1 || !a->children || !s->children)
            return fail(GPK_ERR_INVALID_ARGUMENT, "from_arrow: malformed list level %d", depth);
        int64_t first = 0, last = 0;
        GPK_TRY(rebased_offsets(a->buffers[1], fmt(s, "+L"), lo, hi, level[depth], &first, &last, "list"));
        const ArrowArray* c = a->children[0];
        if (!c || c->offset < 0) return fail(GPK_ERR_INVALID_ARGUMENT, "from_arrow: list level %d without a child", depth);
        lo = c->offset + first;
        hi = c->offset + last;
        a = c;
        s = s->children[0];
        ++depth;
    }
    const int64_t n_coords = hi - lo;
    gpk_geoarrow_desc d;
    memset(&d, 0, sizeof d);
    d.mem_space = GPK_MEM_HOST;
    d.n_geoms = n;
    d.n_coords = n_coords;
    d.validity = validity;
    if (fmt(s, "+s")) {  // Struct<x: f64, y: f64>
        if (s->n_children != 2 || a->n_children != 2 || !fmt(s->children[0], "g") || !fmt(s->children[1], "g"))
            return fail(GPK_ERR_MISMATCHED_GEOMETRY, "from_arrow: coordinates must be Struct<x: f64, y: f64> (two float64 fields)");
        const ArrowArray *cx = a->children[0], *cy = a->children[1];
        if (!cx || !cy || cx->n_buffers < 2 || cy->n_buffers < 2 || cx->offset < 0 || cy->offset < 0)
            return fail(GPK_ERR_INVALID_ARGUMENT, "from_arrow: malformed coordinate fields");
        const double *xb = (const double*)cx->buffers[1], *yb = (const double*)cy->buffers[1];
        if (n_coords > 0 && (!xb || !yb)) return fail(GPK_ERR_INVALID_ARGUMENT, "from_arrow: coordinate buffers are NULL");
        d.x = xb ? xb + cx->offset + lo : nullptr;
        d.y = yb ? yb + cy->offset + lo : nullptr;
    } else if (fmt(s, "+w:2")) {  // FixedSizeList<f64, 2>: interleaved
        if (s->n_children != 1 || a->n_children != 1 || !fmt(s->children[0], "g"))
            return fail(GPK_ERR_MISMATCHED_GEOMETRY, "from_arrow: FixedSizeList coordinates must hold float64");
        const ArrowArray* cv = a->children[0];
        if (!cv || cv->n_buffers < 2 || cv->offset < 0) return fail(GPK_ERR_INVALID_ARGUMENT, "from_arrow: malformed coordinate values");
        const double* vb = (const double*)cv->buffers[1];
        if (n_coords > 0 && !vb) return fail(GPK_ERR_INVALID_ARGUMENT, "from_arrow: coordinate buffer is NULL");
        d.xy = vb ? vb + cv->offset + 2 * lo : nullptr;
    } else {
        return fail(GPK_ERR_MISMATCHED_GEOMETRY, "from_arrow: unsupported column format \"%s\" (WKB binary, or lists over Struct<x, y> / FixedSizeList<f64, 2>)", s->format ? s->format : "");
    }
    // depth 1 and 2 are two geometry types each: the extension name (geoarrow.multipoint, ...) or the caller's hint decides
    int t = -1;
    const bool multi_named = ext == "geoarrow.multipoint" || ext == "geoarrow.multilinestring";
    switch (depth) {
    case 0: t = GPK_GEOM_POINT; break;
    case 1: t = (geom_type_hint == GPK_GEOM_MULTIPOINT || multi_named) ? GPK_GEOM_MULTIPOINT : GPK_GEOM_LINESTRING; break;
    case 2: t = (geom_type_hint == GPK_GEOM_MULTILINESTRING || multi_named) ? GPK_GEOM_MULTILINESTRING : GPK_GEOM_POLYGON; break;
    default: t = GPK_GEOM_MULTIPOLYGON; break;
    }
    if (geom_type_hint >= 0 && geom_type_hint != t) {
        const bool ok = (depth == 1 && (geom_type_hint == GPK_GEOM_LINESTRING || geom_type_hint == GPK_GEOM_MULTIPOINT)) ||
                        (depth == 2 && (geom_type_hint == GPK_GEOM_POLYGON || geom_type_hint == GPK_GEOM_MULTILINESTRING));
        if (!ok) return fail(GPK_ERR_MISMATCHED_GEOMETRY, "from_arrow: a column of %d list level(s) cannot be geometry type %d", depth, geom_type_hint);
        t = geom_type_hint;
    }
    d.geom_type = t;
    if (depth == 1) {
        d.geom_offsets = level[0].data();
    } else if (depth == 2) {
        d.geom_offsets = level[0].data();
        d.ring_offsets = level[1].data();
        d.n_rings = (int64_t)level[1].size() - 1;
    } else if (depth == 3) {
        d.geom_offsets = level[0].data();
        d.part_offsets = level[1].data();
        d.ring_offsets = level[2].data();
        d.n_parts = (int64_t)level[1].size() - 1;
        d.n_rings = (int64_t)level[2].size() - 1;
    }
    if (out_geom_type) *out_geom_type = t;
    return gpk_geoarray_upload(&d, stream, out);  // (host buffers: copied — and a Struct<x, y> column interleaved — before it returns)
}

// ---- the return half of the seam: a handle -> ArrowArray / ArrowSchema with callee-owned buffers and release callbacks ------------------
// (py-geopolars/src/ffi.rs:35-52: `to_py_array` exports every result as such a pair and the importer calls `release` when it is done)
namespace {

struct OwnedArray {  // private_data of an exported ArrowArray: everything its release callback frees
    std::vector<void*> buffers;        // malloc'ed (or NULL), in Arrow's buffer order
    std::vector<ArrowArray*> children; // heap-allocated child structs (released, then freed, with the parent)
    std::vector<const void*> buffer_ptrs;
};
void release_array(ArrowArray* a) {
    if (!a || !a->release) return;
    OwnedArray* o = static_cast<OwnedArray*>(a->private_data);
    for (ArrowArray* c : o->children) {
        if (c->release) c->release(c);
        delete c;
    }
    for (void* b : o->buffers) free(b);
    delete o;
    a->release = nullptr;
    a->private_data = nullptr;
}
// takes ownership of `buffers` and `children`
ArrowArray* make_array(int64_t length, int64_t null_count, std::vector<void*> buffers, std::vector<ArrowArray*> children) {
    OwnedArray* o = new OwnedArray();
    o->buffers = std::move(buffers);
    o->children = std::move(children);
    for (void* b : o->buffers) o->buffer_ptrs.push_back(b);
    ArrowArray* a = new ArrowArray();
    memset(a, 0, sizeof *a);
    a->length = length;
    a->null_count = null_count;
    a->n_buffers = (int64_t)o->buffer_ptrs.size();
    a->buffers = o->buffer_ptrs.data();
    a->n_children = (int64_t)o->children.size();
    a->children = o->children.empty() ? nullptr : o->children.data();
    a->release = release_array;
    a->private_data = o;
    return a;
}
struct OwnedSchema {
    std::string format, name, metadata;
    std::vector<ArrowSchema*> children;
};
void release_schema(ArrowSchema* s) {
    if (!s || !s->release) return;
    OwnedSchema* o = static_cast<OwnedSchema*>(s->private_data);
    for (ArrowSchema* c : o->children) {
        if (c->release) c->release(c);
        delete c;
    }
    delete o;
    s->release = nullptr;
    s->private_data = nullptr;
}
ArrowSchema* make_schema(const char* format, const char* name, const std::string& extension, bool nullable, std::vector<ArrowSchema*> children) {
    OwnedSchema* o = new OwnedSchema();
    o->format = format;
    o->name = name;
    if (!extension.empty()) {  // int32 n = 1, then (int32 len, key, int32 len, value), native endian
        const char key[] = "ARROW:extension:name";
        const int32_t one = 1, kl = (int32_t)(sizeof key - 1), vl = (int32_t)extension.size();
        o->metadata.append((const char*)&one, 4).append((const char*)&kl, 4).append(key, (size_t)kl).append((const char*)&vl, 4).append(extension);
    }
    o->children = std::move(children);
    ArrowSchema* s = new ArrowSchema();
    memset(s, 0, sizeof *s);
    s->format = o->format.c_str();
    s->name = o->name.c_str();
    s->metadata = o->metadata.empty() ? nullptr : o->metadata.data();
    s->flags = nullable ? 2 /* ARROW_FLAG_NULLABLE */ : 0;
    s->n_children = (int64_t)o->children.size();
    s->children = o->children.empty() ? nullptr : o->children.data();
    s->release = release_schema;
    s->private_data = o;
    return s;
}
void* copy_of(const void* src, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (p && bytes) memcpy(p, src, bytes);
    return p;
}

}  // namespace

extern "C" int32_t gpk_geoarray_to_arrow(const gpk_geoarray* a, int32_t layout, void* stream, struct ArrowArray* out_array, struct ArrowSchema* out_schema) {
    if (!a || !out_array || !out_schema) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    if (layout != GPK_ARROW_WKB && layout != GPK_ARROW_INTERLEAVED && layout != GPK_ARROW_STRUCT)
        return fail(GPK_ERR_INVALID_ARGUMENT, "to_arrow: unknown layout %d", layout);
    const int32_t t = a->d.type;
    int64_t sizes[4] = {0, 0, 0, 0};
    GPK_TRY(gpk_geoarray_download(a, sizes, nullptr, nullptr, nullptr, nullptr, stream));
    const int64_t n_coords = sizes[0], n_parts = sizes[1], n_rings = sizes[2], n = sizes[3];
    // validity (the bitmap may live on the device only: a decoded or exploded column)
    std::vector<uint8_t> bits((size_t)((n + 7) / 8), 0);
    int32_t has_validity = 0;
    GPK_TRY(gpk_geoarray_validity(a, bits.empty() ? nullptr : bits.data(), &has_validity, stream));
    int64_t nulls = 0;
    void* validity = nullptr;
    if (has_validity) {
        for (int64_t i = 0; i < n; ++i) nulls += !((bits[(size_t)(i >> 3)] >> (i & 7)) & 1);
        validity = copy_of(bits.data(), bits.size());
    }
    if (layout == GPK_ARROW_WKB) {  // Binary: validity | i32 offsets | values, the bytes encoded on the GPU
        std::vector<int32_t> off((size_t)n + 1, 0);
        int64_t n_bytes = 0;
        const int32_t rc0 = gpk_geoarray_to_wkb(a, off.data(), nullptr, 0, &n_bytes, GPK_MEM_HOST, stream);
        if (rc0 != GPK_OK) {
            free(validity);
            return rc0;
        }
        void* values = malloc(n_bytes ? (size_t)n_bytes : 1);
        if (!values) {
            free(validity);
            return fail(GPK_ERR_OOM, "to_arrow: %lld bytes of WKB", (long long)n_bytes);
        }
        if (n_bytes > 0) {
            const int32_t rc = gpk_geoarray_to_wkb(a, off.data(), (uint8_t*)values, n_bytes, &n_bytes, GPK_MEM_HOST, stream);
            if (rc != GPK_OK) {
                free(values);
                free(validity);
                return rc;
            }
        }
        ArrowArray* arr = make_array(n, nulls, {validity, copy_of(off.data(), sizeof(int32_t) * off.size()), values}, {});
        ArrowSchema* sch = make_schema("z", "geometry", "geoarrow.wkb", true, {});
        *out_array = *arr;
        *out_schema = *sch;
        delete arr;  // (the structs were moved into the caller's: private_data travels with them)
        delete sch;
        return GPK_OK;
    }
    // native GeoArrow: the handle's buffers as they are
    std::vector<double> xy((size_t)n_coords * 2);
    std::vector<int32_t> go(t != GPK_GEOM_POINT ? (size_t)n + 1 : 0), po(t == GPK_GEOM_MULTIPOLYGON ? (size_t)n_parts + 1 : 0),
        ro((t == GPK_GEOM_POLYGON || t == GPK_GEOM_MULTILINESTRING || t == GPK_GEOM_MULTIPOLYGON) ? (size_t)n_rings + 1 : 0);
    const int32_t rc1 = gpk_geoarray_download(a, sizes, xy.empty() ? nullptr : xy.data(), go.empty() ? nullptr : go.data(), po.empty() ? nullptr : po.data(),
                                              ro.empty() ? nullptr : ro.data(), stream);
    if (rc1 != GPK_OK) {
        free(validity);
        return rc1;
    }
    // the coordinate array (length n_coords), innermost
    ArrowArray* arr;
    ArrowSchema* sch;
    const bool point = t == GPK_GEOM_POINT;
    if (layout == GPK_ARROW_STRUCT) {
        double *xs = (double*)malloc(n_coords ? sizeof(double) * (size_t)n_coords : 1), *ys = (double*)malloc(n_coords ? sizeof(double) * (size_t)n_coords : 1);
        if (!xs || !ys) {
            free(xs);
            free(ys);
            free(validity);
            return fail(GPK_ERR_OOM, "to_arrow: coordinates");
        }
        for (int64_t i = 0; i < n_coords; ++i) {
            xs[i] = xy[(size_t)(2 * i)];
            ys[i] = xy[(size_t)(2 * i + 1)];
        }
        ArrowArray* ax = make_array(n_coords, 0, {nullptr, xs}, {});
        ArrowArray* ay = make_array(n_coords, 0, {nullptr, ys}, {});
        arr = make_array(n_coords, point ? nulls : 0, {point ? validity : nullptr}, {ax, ay});
        sch = make_schema("+s", point ? "geometry" : "vertices", point ? "geoarrow.point" : "", point, {make_schema("g", "x", "", false, {}), make_schema("g", "y", "", false, {})});
    } else {
        ArrowArray* av = make_array(n_coords * 2, 0, {nullptr, copy_of(xy.data(), sizeof(double) * xy.size())}, {});
        arr = make_array(n_coords, point ? nulls : 0, {point ? validity : nullptr}, {av});
        sch = make_schema("+w:2", point ? "geometry" : "vertices", point ? "geoarrow.point" : "", point, {make_schema("g", "xy", "", false, {})});
    }
    // list levels, innermost first; the outermost carries the validity bitmap and the extension name
    struct Level {
        std::vector<int32_t>* off;
        int64_t length;
        const char* child_name;
    };
    std::vector<Level> levels;  // innermost .. outermost
    const char* ext = "geoarrow.point";
    switch (t) {
    case GPK_GEOM_POINT: break;
    case GPK_GEOM_LINESTRING: levels = {{&go, n, "vertices"}}; ext = "geoarrow.linestring"; break;
    case GPK_GEOM_MULTIPOINT: levels = {{&go, n, "points"}}; ext = "geoarrow.multipoint"; break;
    case GPK_GEOM_POLYGON: levels = {{&ro, n_rings, "vertices"}, {&go, n, "rings"}}; ext = "geoarrow.polygon"; break;
    case GPK_GEOM_MULTILINESTRING: levels = {{&ro, n_rings, "vertices"}, {&go, n, "linestrings"}}; ext = "geoarrow.multilinestring"; break;
    case GPK_GEOM_MULTIPOLYGON: levels = {{&ro, n_rings, "vertices"}, {&po, n_parts, "rings"}, {&go, n, "polygons"}}; ext = "geoarrow.multipolygon"; break;
    default:
        release_array(arr);
        release_schema(sch);
        delete arr;
        delete sch;
        free(validity);
        return fail(GPK_ERR_MISMATCHED_GEOMETRY, "to_arrow: geometry type %d", t);
    }
    for (size_t i = 0; i < levels.size(); ++i) {
        const bool outer = i + 1 == levels.size();
        // (a child's field name is fixed when its schema is made: the coordinate / inner list schemas are renamed to what this level calls them)
        static_cast<OwnedSchema*>(sch->private_data)->name = levels[i].child_name;
        sch->name = static_cast<OwnedSchema*>(sch->private_data)->name.c_str();
        arr = make_array(levels[i].length, outer ? nulls : 0, {outer ? validity : nullptr, copy_of(levels[i].off->data(), sizeof(int32_t) * levels[i].off->size())}, {arr});
        sch = make_schema("+l", outer ? "geometry" : "", outer ? ext : "", outer, {sch});
    }
    *out_array = *arr;
    *out_schema = *sch;
    delete arr;
    delete sch;
    return GPK_OK;
}
