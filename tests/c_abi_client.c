/* Plain-C client of include/geopolars_hip.h (test infrastructure): proves that the header is valid C99, that the
 * library links from C without any C++ or HIP header, and that compute entry points fail loudly — with a status code
 * and a message, never a crash — when no gfx950 device is present. */
#include <stdio.h>
#include <string.h>

#include "geopolars_hip.h"

int main(void) {
    char msg[256];
    double xy[8] = {0, 0, 1, 0, 1, 1, 0, 0};
    int32_t ring_off[2] = {0, 4}, geom_off[2] = {0, 1};
    gpk_geoarrow_desc d;
    gpk_geoarray* h = NULL;
    int32_t n_dev = -1, rc;

    if (!gpk_version() || !strstr(gpk_version(), "geopolars_hip")) return 10;
    memset(&d, 0, sizeof d);
    d.geom_type = GPK_GEOM_POLYGON;
    d.mem_space = GPK_MEM_HOST;
    d.n_geoms = 1;
    d.n_coords = 4;
    d.n_rings = 1;
    d.xy = xy;
    d.geom_offsets = geom_off;
    d.ring_offsets = ring_off;
    rc = gpk_device_count(&n_dev);
    printf("devices: rc=%d n=%d\n", (int)rc, (int)n_dev);
    rc = gpk_geoarray_upload(&d, NULL, &h);
    if (rc == GPK_OK) { /* a GPU is present: one operator, then the column out and in again through the C Data Interface structs */
        double area = -1.0;
        struct ArrowArray arr;
        struct ArrowSchema sch;
        gpk_geoarray* back = NULL;
        int32_t gt = -1;
        rc = gpk_area(h, &area, GPK_MEM_HOST, NULL);
        printf("area: rc=%d value=%g\n", (int)rc, area);
        if (rc != GPK_OK || area != 0.5) return 11;
        memset(&arr, 0, sizeof arr);
        memset(&sch, 0, sizeof sch);
        rc = gpk_geoarray_to_arrow(h, GPK_ARROW_STRUCT, NULL, &arr, &sch);
        if (rc != GPK_OK || arr.length != 1 || arr.n_children != 1 || !arr.release || !sch.release || strcmp(sch.format, "+l") != 0) return 15;
        rc = gpk_geoarray_from_arrow(&arr, &sch, -1, NULL, &back, &gt); /* (borrowed: still ours to release) */
        if (rc != GPK_OK || gt != GPK_GEOM_POLYGON) return 16;
        area = -1.0;
        rc = gpk_area(back, &area, GPK_MEM_HOST, NULL);
        printf("area after the round trip: rc=%d value=%g\n", (int)rc, area);
        arr.release(&arr);
        sch.release(&sch);
        if (arr.release || sch.release) return 17; /* a released struct says so */
        gpk_geoarray_free(back);
        gpk_geoarray_free(h);
        return (rc == GPK_OK && area == 0.5) ? 0 : 18;
    }
    gpk_last_error(msg, sizeof msg);
    printf("no device: rc=%d message=\"%s\"\n", (int)rc, msg);
    if (rc != GPK_ERR_DEVICE || msg[0] == 0) return 12;
    /* argument validation does not need a device */
    if (gpk_geoarray_upload(NULL, NULL, &h) != GPK_ERR_INVALID_ARGUMENT) return 13;
    {
        int64_t n_bytes = -1;
        d.mem_space = GPK_MEM_HOST;
        if (gpk_wkb_encode(&d, NULL, NULL, 0, &n_bytes) != GPK_OK || n_bytes != 1 + 4 + 4 + 4 + 4 * 16) return 14; /* host encoder */
    }
    return 0;
}
