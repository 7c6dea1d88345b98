/*
 * rt_oracle.c -- CPU restatement of the MasonSchuckman/Raytracer render path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the checker that the HIP path is
 * compared against (tests/, the smoke test and the cpu_baseline leg of the
 * benchmark).  Nothing in the product library links, loads or calls it.
 *
 * It restates, in plain C and in the reference's own recursive form, what the
 * reference computes for one frame: the .nff dialect, the ear clipping of
 * polygons and patches, the loose bounding-sphere octree, closest-hit and
 * shadow traversal, the ray tree, Phong shading and the light-camera caustics.
 * Every function cites the reference file:line it follows.
 *
 * Vector arithmetic: the reference uses Eigen 3.4.0 (header-only, not vendored,
 * not installed here).  Its fixed-size 3-vector operations are restated as:
 *   dot / squaredNorm : (x0 + x1) + x2       (Eigen's unrolled packet redux)
 *   norm              : sqrt(squaredNorm)
 *   normalized        : v / sqrt(v.v) when v.v > 0, else v unchanged
 *   cross             : the textbook component formula
 *   3x3 determinant   : Eigen's bruteforce_det3_helper expansion
 * Scalar * vector and vector / scalar are component-wise IEEE operations.
 *
 * Pinning: the reference holds one golden artefact, hide.ppm (its render of
 * glass.nff); the oracle's render of glass.nff is compared byte-for-byte with
 * it (tests/test_oracle_golden.py).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no -ffast-math).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ vec3 */
typedef struct { double x, y, z; } v3;

static inline v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline v3 vmul(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }   /* v * s  */
static inline v3 smul(double s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }   /* s * v  */
static inline v3 vdiv(v3 a, double s) { return V(a.x / s, a.y / s, a.z / s); }
static inline double vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline double vsq(v3 a) { return vdot(a, a); }
static inline double vnorm(v3 a) { return sqrt(vsq(a)); }
static inline v3 vnormalized(v3 a)
{
    double z = vsq(a);
    if (z > 0) return vdiv(a, sqrt(z));
    return a;
}
static inline v3 vcross(v3 a, v3 b)
{
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline double vget(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

/* Eigen 3.4 determinant of the 3x3 matrix whose COLUMNS are a, b, c. */
static double eigen_det3_cols(v3 a, v3 b, v3 c)
{
    double h012 = a.x * (b.y * c.z - c.y * b.z);
    double h102 = b.x * (a.y * c.z - c.y * a.z);
    double h201 = c.x * (a.y * b.z - b.y * a.z);
    return (h012 - h102) + h201;
}

/* ------------------------------------------------------------- scene types */
typedef struct { double r, g, b, Kd, Ks, Shine, T, ior; } Material;   /* Solid.h:27-36 */

enum { SOLID_TRI = 0, SOLID_SPHERE = 1 };                              /* type(): Triangle.h:381, Sphere.h:371 */

typedef struct {
    int type;
    Material m;
    /* triangle (Triangle.h:12-46) */
    v3 v[3];
    v3 ab, ac, normal;
    int has_vn;
    v3 vn[3];
    /* sphere (Sphere.h:19-31) */
    v3 pos;
    double radius;
} Solid;

typedef struct { v3 pos; v3 color; double intensity; } Light;          /* Light.h:6-20 */

typedef struct Node {                                                  /* Octree.h:733-907 */
    v3 origin;
    double size;
    double brad;
    struct Node *child[8];
    int *solids;
    int nsolids;
} Node;

/* point octree of the light camera, Octree.h:1347-1497 */
typedef struct PNode {
    v3 origin;
    double size;
    int leaf;
    int begin, end;         /* this node's points: perm[begin..end) (allChildObs) */
    struct PNode *child[8];
} PNode;

typedef struct {                                                        /* LightCam.h:13-65 */
    int sphere, light;
    double spw;
    v3 light_pos, sphere_pos;
    int npts;
    v3 *pts;
    double *w;
    int *perm;
    PNode *root;
    double maxDistFromZero;
    uint64_t light_rays;
} LightCam;

typedef struct {
    /* nff globals, Driver.cpp:30-44 */
    v3 bg, from, at, up;
    double angle, hither;
    int resx, resy;
    int maxReflectionsQ;      /* parsed, unused (Driver.cpp:501-504) */
    int lightCamSamples;      /* Z, default 150 (Driver.cpp:41) */
    Solid *solids; int nsolids, csolids;
    Light *lights; int nlights, clights;
    int phong;
    Node *root; int nnodes;
    /* light cameras, created lazily like Camera.h:417-440 */
    LightCam **cams; int ncams, ccams;
    pthread_mutex_t cam_lock;
} Scene;

/* camera, Camera.h:19-46 */
typedef struct {
    v3 u, v, w, e, bg;
    double d, distToEdge, pixWidth, l, t;
    int resx, resy, numSamples;
    double aperture;
} Camera;

typedef struct {                                                        /* Solid.h:13-25 */
    int nearest;
    double t;
    v3 normal, rayDir, e, impactPos, reflectionDir;
} HitRecord;

typedef struct {
    uint64_t closest_rays, shadow_rays, node_tests_closest, prim_tests_closest,
             node_tests_shadow, prim_tests_shadow, lightcam_queries;
} Counters;

/* ---------------------------------------------------------------- parsing */
static void push_solid(Scene *s, const Solid *x)
{
    if (s->nsolids == s->csolids) {
        s->csolids = s->csolids ? s->csolids * 2 : 256;
        s->solids = (Solid *)realloc(s->solids, sizeof(Solid) * (size_t)s->csolids);
    }
    s->solids[s->nsolids++] = *x;
}

/* Triangle::finalize, Triangle.h:40-46 */
static void tri_finalize(Solid *t)
{
    t->ab = vsub(t->v[0], t->v[1]);
    t->ac = vsub(t->v[0], t->v[2]);
    t->normal = vnormalized(vcross(t->ab, t->ac));
}

static void add_triangle(Scene *s, const Material *m, v3 a, v3 b, v3 c, const v3 *na, const v3 *nb, const v3 *nc)
{
    Solid t;
    memset(&t, 0, sizeof t);
    t.type = SOLID_TRI;
    t.m = *m;
    t.v[0] = a; t.v[1] = b; t.v[2] = c;
    if (na) { t.has_vn = 1; t.vn[0] = *na; t.vn[1] = *nb; t.vn[2] = *nc; }
    tri_finalize(&t);
    push_solid(s, &t);
}

/* Polygon::checkPointInTriangle, Polygon.h:97-105 (z component of the crosses only) */
static int point_in_tri_z(v3 a, v3 b, v3 c, v3 q)
{
    double s1 = vcross(vsub(a, b), vsub(a, q)).z;
    double s2 = vcross(vsub(b, c), vsub(b, q)).z;
    double s3 = vcross(vsub(c, a), vsub(c, q)).z;
    return (s1 > 0 && s2 > 0 && s3 > 0) || (s1 < 0 && s2 < 0 && s3 < 0);
}

/* Polygon::naiveEarClipping Polygon.h:107-205 / PolygonalPatch::naiveEarClipping PolygonalPatch.h:78-157 */
static void ear_clip(Scene *s, const Material *m, const v3 *verts, const v3 *norms, int n)
{
    v3 normal_ = vnormalized(vcross(vsub(verts[0], verts[1]), vsub(verts[2], verts[1])));
    v3 *v = (v3 *)malloc(sizeof(v3) * (size_t)n);
    v3 *nn = norms ? (v3 *)malloc(sizeof(v3) * (size_t)n) : NULL;
    int vs = n;
    memcpy(v, verts, sizeof(v3) * (size_t)n);
    if (norms) memcpy(nn, norms, sizeof(v3) * (size_t)n);
    double deg = 180.0 / 3.14159;
    for (int i = 0; i < vs - 3; i++) {
        int ia = (i - 1 + vs) % vs, ib = i, ic = (i + 1) % vs;
        v3 a = v[ia], b = v[ib], c = v[ic];
        v3 ba = vsub(b, a), bc = vsub(b, c);
        double angle = atan2(eigen_det3_cols(ba, bc, normal_), vdot(ba, bc));
        angle *= deg;
        if (angle > 180 || angle < 0) continue;
        int ear = 1;
        for (int test = 0; test < vs; test++) {
            if (test != ia && test != ib && test != ic)
                if (point_in_tri_z(a, b, c, v[test])) { ear = 0; break; }
        }
        if (!ear) continue;
        if (nn) add_triangle(s, m, a, b, c, &nn[ia], &nn[ib], &nn[ic]);
        else add_triangle(s, m, a, b, c, NULL, NULL, NULL);
        memmove(v + i, v + i + 1, sizeof(v3) * (size_t)(vs - i - 1));
        if (nn) memmove(nn + i, nn + i + 1, sizeof(v3) * (size_t)(vs - i - 1));
        vs--;
        i = -1;
    }
    if (nn) add_triangle(s, m, v[0], v[1], v[2], &nn[0], &nn[1], &nn[2]);
    else add_triangle(s, m, v[0], v[1], v[2], NULL, NULL, NULL);
    free(v);
    free(nn);
}

/* token reader over one line: skips blanks, returns token start or NULL; advances *p */
static const char *next_tok(const char **p, int *len)
{
    const char *q = *p;
    while (*q == ' ' || *q == '\t' || *q == '\r' || *q == '\n' || *q == '\v' || *q == '\f') q++;
    if (!*q) { *p = q; return NULL; }
    const char *st = q;
    while (*q && !(*q == ' ' || *q == '\t' || *q == '\r' || *q == '\n' || *q == '\v' || *q == '\f')) q++;
    *len = (int)(q - st);
    *p = q;
    return st;
}
static int read_double(const char **p, double *out)
{
    const char *q = *p;
    char *end;
    while (*q == ' ' || *q == '\t' || *q == '\r' || *q == '\v' || *q == '\f') q++;
    if (!*q) return 0;
    double d = strtod(q, &end);
    if (end == q) return 0;
    *out = d;
    *p = end;
    return 1;
}
static int read_int(const char **p, int *out)
{
    const char *q = *p;
    char *end;
    while (*q == ' ' || *q == '\t' || *q == '\r' || *q == '\v' || *q == '\f') q++;
    if (!*q) return 0;
    long d = strtol(q, &end, 10);
    if (end == q) return 0;
    *out = (int)d;
    *p = end;
    return 1;
}
static void skip_word(const char **p) { int l; (void)next_tok(p, &l); }

typedef struct { char **line; int n; } Lines;
static Lines split_lines(char *buf)
{
    Lines L = {NULL, 0};
    int cap = 0;
    char *p = buf;
    while (*p) {
        if (L.n == cap) { cap = cap ? cap * 2 : 1024; L.line = (char **)realloc(L.line, sizeof(char *) * (size_t)cap); }
        L.line[L.n++] = p;
        char *nl = strchr(p, '\n');
        if (!nl) break;
        *nl = 0;
        p = nl + 1;
    }
    return L;
}

/* parseInput, Driver.cpp:426-691.  The material persists from one `f` line to
 * the next (the reference declares it inside the loop, Driver.cpp:445, which is
 * undefined behaviour; the persisting reading is the one its scenes rely on). */
static int parse_nff(Scene *s, const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)sz + 1);
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); return 0; }
    buf[sz] = 0;
    fclose(f);
    Lines L = split_lines(buf);
    Material props;
    memset(&props, 0, sizeof props);
    static const char empty[] = "";
#define NEXTLINE() ((li + 1 < L.n) ? L.line[++li] : (li = L.n, empty))
    for (int li = 0; li < L.n; li++) {
        const char *line = L.line[li];
        const char *p = line;
        if (line[0] == 'p' && line[1] == 'p') {                       /* Driver.cpp:448-490 */
            int nv = 0;
            skip_word(&p);
            read_int(&p, &nv);
            v3 *vv = (v3 *)calloc((size_t)(nv > 0 ? nv : 1), sizeof(v3));
            v3 *nn = (v3 *)calloc((size_t)(nv > 0 ? nv : 1), sizeof(v3));
            v3 vertex = V(0, 0, 0), normal = V(0, 0, 0);
            for (int k = 0; k < nv; k++) {
                const char *q = NEXTLINE();
                read_double(&q, &vertex.x) && read_double(&q, &vertex.y) && read_double(&q, &vertex.z) &&
                    read_double(&q, &normal.x) && read_double(&q, &normal.y) && read_double(&q, &normal.z);
                vv[k] = vertex;
                nn[k] = vnormalized(normal);                           /* PolygonalPatch.h:32-36 */
            }
            if (nv >= 3) ear_clip(s, &props, vv, nn, nv);
            free(vv);
            free(nn);
            continue;
        }
        switch (line[0]) {
        case 'b':
            skip_word(&p);
            read_double(&p, &s->bg.x) && read_double(&p, &s->bg.y) && read_double(&p, &s->bg.z);
            break;
        case 'Q':
            skip_word(&p);
            read_int(&p, &s->maxReflectionsQ);
            break;
        case 'Z':
            skip_word(&p);
            read_int(&p, &s->lightCamSamples);
            break;
        case 'v': {                                                    /* Driver.cpp:509-546 */
            const char *q;
            q = NEXTLINE(); skip_word(&q);
            read_double(&q, &s->from.x) && read_double(&q, &s->from.y) && read_double(&q, &s->from.z);
            q = NEXTLINE(); skip_word(&q);
            read_double(&q, &s->at.x) && read_double(&q, &s->at.y) && read_double(&q, &s->at.z);
            q = NEXTLINE(); skip_word(&q);
            read_double(&q, &s->up.x) && read_double(&q, &s->up.y) && read_double(&q, &s->up.z);
            q = NEXTLINE(); skip_word(&q);
            read_double(&q, &s->angle);
            q = NEXTLINE(); skip_word(&q);
            read_double(&q, &s->hither);
            q = NEXTLINE(); skip_word(&q);
            read_int(&q, &s->resx) && read_int(&q, &s->resy);
            break;
        }
        case 'l': {                                                    /* Driver.cpp:547-567 */
            Light lt;
            lt.color = V(1, 1, 1);
            lt.pos = V(0, 0, 0);
            lt.intensity = 1;
            skip_word(&p);
            read_double(&p, &lt.pos.x) && read_double(&p, &lt.pos.y) && read_double(&p, &lt.pos.z);
            /* the reference reads a colour only when the stream is not at EOF,
             * skipping one token first (Driver.cpp:555-561); no shipped scene has one */
            if (*p) {
                v3 c;
                const char *q = p;
                skip_word(&q);
                if (read_double(&q, &c.x) && read_double(&q, &c.y) && read_double(&q, &c.z)) lt.color = c;
            }
            if (s->nlights == s->clights) {
                s->clights = s->clights ? s->clights * 2 : 8;
                s->lights = (Light *)realloc(s->lights, sizeof(Light) * (size_t)s->clights);
            }
            s->lights[s->nlights++] = lt;
            break;
        }
        case 'T': {                                                    /* Driver.cpp:568-597 */
            v3 vv[3];
            v3 vertex = V(0, 0, 0);
            for (int k = 0; k < 3; k++) {
                const char *q = NEXTLINE();
                read_double(&q, &vertex.x) && read_double(&q, &vertex.y) && read_double(&q, &vertex.z);
                vv[k] = vertex;
            }
            add_triangle(s, &props, vv[0], vv[1], vv[2], NULL, NULL, NULL);
            break;
        }
        case 'p': {                                                    /* Driver.cpp:598-636 */
            int nv = 0;
            skip_word(&p);
            read_int(&p, &nv);
            v3 *vv = (v3 *)calloc((size_t)(nv > 0 ? nv : 1), sizeof(v3));
            v3 vertex = V(0, 0, 0);
            for (int k = 0; k < nv; k++) {
                const char *q = NEXTLINE();
                read_double(&q, &vertex.x) && read_double(&q, &vertex.y) && read_double(&q, &vertex.z);
                vv[k] = vertex;
            }
            if (nv >= 3) ear_clip(s, &props, vv, NULL, nv);
            free(vv);
            break;
        }
        case 'f': {                                                    /* Driver.cpp:637-644 */
            double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            skip_word(&p);
            for (int k = 0; k < 8; k++) if (!read_double(&p, &a[k])) break;
            props.r = a[0]; props.g = a[1]; props.b = a[2]; props.Kd = a[3];
            props.Ks = a[4]; props.Shine = a[5]; props.T = a[6]; props.ior = a[7];
            break;
        }
        case 's': {                                                    /* Driver.cpp:645-661 */
            Solid sp;
            memset(&sp, 0, sizeof sp);
            sp.type = SOLID_SPHERE;
            sp.m = props;
            skip_word(&p);
            read_double(&p, &sp.pos.x) && read_double(&p, &sp.pos.y) && read_double(&p, &sp.pos.z) &&
                read_double(&p, &sp.radius);
            push_solid(s, &sp);
            break;
        }
        default:
            break;
        }
    }
#undef NEXTLINE
    for (int i = 0; i < s->nlights; i++) s->lights[i].intensity = 1.0 / sqrt((double)s->nlights);  /* Driver.cpp:685-688 */
    free(L.line);
    free(buf);
    return 1;
}

/* ------------------------------------------------------------ octree build */
/* Triangle::insideOctree Triangle.h:359-374, Sphere::insideOctree Sphere.h:211-232 */
static int solid_inside(const Solid *s, v3 c1, v3 c2)
{
    if (s->type == SOLID_TRI) {
        int in = 1;
        for (int i = 0; i < 3; i++)
            for (int d = 0; d < 3; d++)
                in = in && (vget(c1, d) <= vget(s->v[i], d) && vget(s->v[i], d) <= vget(c2, d));
        return in;
    }
    double radSquared = s->radius * s->radius;
    double minimumDistance = 0;
    for (int d = 0; d < 3; d++) {
        double p = vget(s->pos, d), a = vget(c1, d), b = vget(c2, d);
        if (a > p) minimumDistance += (p - a) * (p - a);
        else if (b < p) minimumDistance += (p - b) * (p - b);
    }
    if (minimumDistance >= radSquared) return 0;
    return 1;
}

/* setupOctantBounds + makeBounds, Octree.h:1282-1320 */
static void octant_bounds(v3 origin, double size, v3 c1[8], v3 c2[8])
{
    double x = origin.x, y = origin.y, z = origin.z, h = size / 2;
    c1[0] = V(x - size / 2, y - size / 2, z - size / 2);
    c1[1] = V(x, y - size / 2, z - size / 2);
    c1[2] = V(x - size / 2, y, z - size / 2);
    c1[3] = V(x, y, z - size / 2);
    c1[4] = V(x - size / 2, y - size / 2, z);
    c1[5] = V(x, y - size / 2, z);
    c1[6] = V(x - size / 2, y, z);
    c1[7] = V(x, y, z);
    for (int i = 0; i < 8; i++) c2[i] = V(c1[i].x + h, c1[i].y + h, c1[i].z + h);
}

/* Octree<Solid*> constructor, Octree.h:780-907.  Lists keep ascending solid index. */
static Node *build_node(Scene *s, v3 roots, double size, const int *objs, int n)
{
    Node *nd = (Node *)calloc(1, sizeof(Node));
    s->nnodes++;
    nd->size = size;
    nd->origin = V(roots.x + size / 2, roots.y + size / 2, roots.z + size / 2);
    nd->brad = (size * sqrt(3)) / 2.0;
    v3 c1[8], c2[8];
    octant_bounds(nd->origin, size, c1, c2);
    const double MIN_SIZE = .01;
    const int minObs = 5;
    int *lists[8] = {0};
    int cnt[8] = {0};
    char *rem = (char *)calloc((size_t)(n > 0 ? n : 1), 1);
    if (size / 2 > MIN_SIZE && n > minObs) {
        for (int i = 0; i < 8; i++) lists[i] = (int *)malloc(sizeof(int) * (size_t)n);
        for (int k = 0; k < n; k++)
            for (int i = 0; i < 8; i++)
                if (solid_inside(&s->solids[objs[k]], c1[i], c2[i])) {
                    rem[k] = 1;
                    lists[i][cnt[i]++] = objs[k];
                }
    }
    for (int i = 0; i < 8; i++)
        if (cnt[i] > 0) nd->child[i] = build_node(s, c1[i], c2[i].x - c1[i].x, lists[i], cnt[i]);
    nd->solids = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int k = 0; k < n; k++) if (!rem[k]) nd->solids[nd->nsolids++] = objs[k];
    for (int i = 0; i < 8; i++) free(lists[i]);
    free(rem);
    return nd;
}

/* setupOctree, Driver.cpp:224-250 */
static void build_octree(Scene *s)
{
    double size = 100;
    int *all = (int *)malloc(sizeof(int) * (size_t)(s->nsolids > 0 ? s->nsolids : 1));
    for (int i = 0; i < s->nsolids; i++) all[i] = i;
    s->nnodes = 0;
    s->root = build_node(s, V(0 - size / 2, 0 - size / 2, 0 - size / 2), size, all, s->nsolids);
    free(all);
}

static void free_node(Node *n)
{
    if (!n) return;
    for (int i = 0; i < 8; i++) free_node(n->child[i]);
    free(n->solids);
    free(n);
}

/* ------------------------------------------------------- primitive tests */
/* Triangle::det, Triangle.h:388-393 */
static inline double tri_det(v3 a, v3 b, v3 c)
{
    return a.x * (b.y * c.z - c.y * b.z) + b.x * (c.y * a.z - a.y * c.z) + c.x * (a.y * b.z - b.y * a.z);
}

/* Solid::updateHitRecord, Solid.h:110-125 */
static inline void update_hit(int id, double t, v3 rayDir, v3 e, v3 normal, HitRecord *hr)
{
    if (t < hr->t) {
        hr->t = t;
        hr->nearest = id;
        hr->rayDir = rayDir;
        hr->e = e;
        hr->normal = vnormalized(normal);
        hr->impactPos = vadd(e, vmul(rayDir, t));
        hr->reflectionDir = vnormalized(vadd(rayDir, smul(2.0 * -vdot(rayDir, hr->normal), hr->normal)));
    }
}

/* Triangle::intersect(...,hr), Triangle.h:91-158 */
static double tri_intersect(const Solid *s, int id, int phong, v3 rayDir, v3 e, double minT, double maxT, HitRecord *hr)
{
    v3 ae = vsub(s->v[0], e);
    double detA = tri_det(s->ab, s->ac, rayDir);
    double alpha = tri_det(ae, s->ac, rayDir) / detA;
    if (alpha < 0 || alpha > 1) return maxT;
    double beta = tri_det(s->ab, ae, rayDir) / detA;
    if (beta < 0 || beta > 1 - alpha) return maxT;
    double t = tri_det(s->ab, s->ac, ae) / detA;
    if (t < minT || t > maxT) return maxT;
    v3 n = s->normal;
    if (s->has_vn && phong)
        n = vadd(vadd(smul(alpha, s->vn[1]), smul(beta, s->vn[2])), smul(1 - (alpha + beta), s->vn[0]));
    update_hit(id, t, rayDir, e, n, hr);
    return t;
}

/* Sphere::intersect(...,hr), Sphere.h:35-77 */
static double sphere_intersect(const Solid *s, int id, v3 rayDir, v3 e, double minT, double maxT, HitRecord *hr)
{
    double ddotemc = vdot(rayDir, vsub(e, s->pos));
    double d2 = vsq(rayDir);
    double disc = (ddotemc * ddotemc) - d2 * (vsq(vsub(e, s->pos)) - s->radius * s->radius);
    if (disc < 0) return maxT;
    double root1 = (-ddotemc + sqrt(disc)) / d2;
    double root2 = (-ddotemc - sqrt(disc)) / d2;
    double t = root1;
    if (root1 < 0 || (root2 > minT && root2 < root1)) t = root2;
    if (t < minT || t > maxT) return maxT;
    v3 normal = vdiv(vsub(vadd(e, smul(t, rayDir)), s->pos), s->radius);
    update_hit(id, t, rayDir, e, normal, hr);
    return t;
}

/* Sphere::intersect 4-argument form used on node bounding spheres, Sphere.h:155-204 */
static double bound_intersect(v3 pos, double radius, v3 rayDir, v3 e, double minT, double maxT)
{
    v3 eMinusPos = vsub(e, pos);
    double C = vdot(eMinusPos, eMinusPos) - radius * radius;
    double A = vsq(rayDir);
    double B = 2.0 * vdot(rayDir, eMinusPos);
    double discriminant = B * B - 4 * A * C;
    if (discriminant < 0) return maxT;
    double b = B / 2;
    double root = sqrt(discriminant);
    double denom = A;
    double t1 = (-b + root) / denom;
    double t2 = (-b - root) / denom;
    if (t2 > minT) return t2;
    else if (t1 > minT) return t1;
    else return maxT;
}

static inline double solid_intersect(const Scene *sc, int id, v3 rayDir, v3 e, double minT, double maxT, HitRecord *hr)
{
    const Solid *s = &sc->solids[id];
    if (s->type == SOLID_TRI) return tri_intersect(s, id, sc->phong, rayDir, e, minT, maxT, hr);
    return sphere_intersect(s, id, rayDir, e, minT, maxT, hr);
}

/* Octree<Solid*>::intersect, Octree.h:996-1062 */
static void octree_intersect(const Scene *sc, const Node *nd, v3 rayDir, v3 e, double minT, double maxT, HitRecord *hr,
                             uint64_t *node_tests, uint64_t *prim_tests)
{
    (*node_tests)++;
    if (bound_intersect(nd->origin, nd->brad, rayDir, e, minT, maxT) >= maxT) return;
    for (int k = 0; k < nd->nsolids; k++) {
        (*prim_tests)++;
        solid_intersect(sc, nd->solids[k], rayDir, e, minT, maxT, hr);
    }
    for (int i = 0; i < 8; i++)
        if (nd->child[i]) octree_intersect(sc, nd->child[i], rayDir, e, minT, maxT, hr, node_tests, prim_tests);
}

/* Octree<Solid*>::checkOccluded3, Octree.h:1198-1264 */
static int check_occluded3(const Scene *sc, const Node *nd, int parent, v3 rayDir, v3 e, double minT, double maxT,
                           int *onlyTransparent, int *closest, uint64_t *node_tests, uint64_t *prim_tests)
{
    int anything = 0;
    (*node_tests)++;
    if (bound_intersect(nd->origin, nd->brad, rayDir, e, minT, 10000000) >= 10000000) return 0;
    HitRecord hr;
    hr.t = maxT;
    hr.nearest = -1;
    double t = maxT;
    double furthest = minT;
    for (int k = 0; k < nd->nsolids; k++) {
        int id = nd->solids[k];
        t = maxT;
        if (parent != id) {
            (*prim_tests)++;
            t = solid_intersect(sc, id, rayDir, e, minT, maxT, &hr);
        }
        if (t < maxT) {
            anything = 1;
            if (t > furthest) { furthest = t; *closest = id; }
            if (sc->solids[id].m.T == 0) { *onlyTransparent = 0; return 1; }
            else if (sc->solids[id].type != SOLID_SPHERE) { *onlyTransparent = 0; return 1; }
        }
    }
    for (int i = 0; i < 8; i++)
        if (nd->child[i] && check_occluded3(sc, nd->child[i], parent, rayDir, e, minT, maxT, onlyTransparent, closest,
                                            node_tests, prim_tests))
            return 1;
    return anything;
}

/* ----------------------------------------------------------------- camera */
/* Camera::computeCoordinateSystem, Camera.h:64-97.  `radians` selects the
 * LightCam variant (LightCam.h:77-115), which takes its angle in radians. */
static void camera_setup(Camera *c, v3 from, v3 at, v3 up, v3 bg, double angle, int resx, int resy, int radians)
{
    c->bg = bg;
    c->w = vnormalized(vsub(from, at));
    c->u = vnormalized(vcross(up, c->w));
    c->v = vnormalized(vcross(c->w, c->u));
    double aspectRatio = (double)(resx / resy);          /* integer division, Camera.h:73 */
    c->d = vnorm(vsub(from, at));
    c->e = from;
    if (radians) c->distToEdge = c->d * tan(angle / 2);
    else c->distToEdge = c->d * tan(angle / 2 * 3.14159 / 180.0);   /* `rad` = PI / 180.0 textual macro, Camera.h:13-14,83 */
    c->pixWidth = 2 * c->distToEdge / resx;
    c->l = -c->distToEdge + 0.5 * c->pixWidth;
    c->t = c->distToEdge * aspectRatio - 0.5 * c->pixWidth;
    c->resx = resx;
    c->resy = resy;
}

/* ------------------------------------------------------- light camera path */
/* Octree<Vector3d*>::insideOctree, Octree.h:1499-1510 */
static inline int point_inside(v3 p, v3 c1, v3 c2)
{
    return (c1.x <= p.x && p.x <= c2.x) && (c1.y <= p.y && p.y <= c2.y) && (c1.z <= p.z && p.z <= c2.z);
}

/* Octree<Vector3d*> constructor, Octree.h:1394-1497.  perm[begin..end) holds this
 * node's points in insertion order; children take stable sub-ranges. */
static PNode *build_pnode(LightCam *lc, v3 roots, double size, int begin, int end, int *scratch, unsigned char *which)
{
    PNode *nd = (PNode *)calloc(1, sizeof(PNode));
    nd->size = size;
    nd->origin = V(roots.x + size / 2, roots.y + size / 2, roots.z + size / 2);
    nd->begin = begin;
    nd->end = end;
    nd->leaf = 1;
    int n = end - begin;
    const double MIN_SIZE = .00000001;
    const int minObs = 10;
    if (!(size / 2 > MIN_SIZE && n > minObs)) return nd;
    v3 c1[8], c2[8];
    octant_bounds(nd->origin, size, c1, c2);
    int cnt[9] = {0};
    for (int k = begin; k < end; k++) {
        v3 p = lc->pts[lc->perm[k]];
        int w = 8;
        for (int i = 0; i < 8; i++)
            if (point_inside(p, c1[i], c2[i])) { w = i; break; }
        which[k] = (unsigned char)w;
        cnt[w]++;
    }
    int numRemoved = n - cnt[8];
    if (numRemoved == n) nd->leaf = 0;
    int off[9], pos[9];
    off[0] = begin;
    for (int i = 1; i < 9; i++) off[i] = off[i - 1] + cnt[i - 1];
    memcpy(pos, off, sizeof pos);
    for (int k = begin; k < end; k++) scratch[pos[which[k]]++] = lc->perm[k];
    memcpy(lc->perm + begin, scratch + begin, sizeof(int) * (size_t)n);
    for (int i = 0; i < 8; i++)
        if (cnt[i] > 0) nd->child[i] = build_pnode(lc, c1[i], c2[i].x - c1[i].x, off[i], off[i] + cnt[i], scratch, which);
    return nd;
}

static void free_pnode(PNode *n)
{
    if (!n) return;
    for (int i = 0; i < 8; i++) free_pnode(n->child[i]);
    free(n);
}

static void lc_push(LightCam *lc, int *cap, v3 p, double w)
{
    if (lc->npts == *cap) {
        *cap = *cap ? *cap * 2 : 4096;
        lc->pts = (v3 *)realloc(lc->pts, sizeof(v3) * (size_t)*cap);
        lc->w = (double *)realloc(lc->w, sizeof(double) * (size_t)*cap);
    }
    lc->pts[lc->npts] = p;
    lc->w[lc->npts] = w;
    lc->npts++;
}

/* LightCam::recursiveCastLightRays + castLightRay, LightCam.h:340-556.
 * Only loop iteration i = 0 is ever live (hr[i+1].nearest is cleared at :479),
 * so the ray "tree" is a refraction/TIR chain. */
static void light_ray(const Scene *sc, LightCam *lc, int *cap, v3 rayDir, v3 e, int maxReflections, double CoR, double coef,
                      int top)
{
    const double maxT = 10000000000.0, minT = .000001;
    HitRecord hr;
    hr.nearest = -1;
    hr.t = maxT;
    uint64_t nt = 0, pt = 0;
    lc->light_rays++;
    if (coef > .05) octree_intersect(sc, sc->root, rayDir, e, minT, maxT, &hr, &nt, &pt);
    if (hr.nearest < 0) return;
    const Solid *hs = &sc->solids[hr.nearest];
    lc_push(lc, cap, hr.impactPos, coef * (1.0 - hs->m.T));                 /* LightCam.h:533-537 */
    double distToOrigin = vnorm(hr.impactPos);
    lc->maxDistFromZero = fmax(distToOrigin, lc->maxDistFromZero);
    if (top && hs->type != SOLID_SPHERE) return;                           /* LightCam.h:360 */
    if (maxReflections <= 0) return;                                        /* loop body never entered */
    double coefOfReflection = CoR * hs->m.Ks;
    if (coef < .01) return;
    if (coefOfReflection < .01 && hs->m.T == 0) return;
    if (hs->m.T > 0) {
        double ior = hs->m.ior;
        v3 normal = hr.normal;
        v3 viewDir = vneg(hr.rayDir);
        double dot = vdot(normal, viewDir);
        int entering = (dot >= 0);
        if (entering) ior = 1.0 / ior;
        else { normal = vneg(normal); dot = -dot; }
        double T = hs->m.T;
        double discriminate = 1 - (1 - (dot * dot)) * (ior * ior);
        if (discriminate > 0) {
            v3 refractionDir = vsub(vmul(vsub(vmul(normal, dot), viewDir), ior), vmul(normal, sqrt(discriminate)));
            refractionDir = vnormalized(refractionDir);
            light_ray(sc, lc, cap, refractionDir, hr.impactPos, maxReflections - 1, coefOfReflection, coef * T, 0);
        } else {
            v3 ird = vnormalized(vadd(hr.rayDir, smul(2.0 * -vdot(hr.rayDir, normal), normal)));
            light_ray(sc, lc, cap, ird, hr.impactPos, maxReflections - 1, coefOfReflection, coef * T, 0);
        }
    }
}

/* Camera.h:417-440 + LightCam::lightCast LightCam.h:207-306 + setupLightOctree :308-337 */
static LightCam *lightcam_create(const Scene *sc, int sphere, int light)
{
    LightCam *lc = (LightCam *)calloc(1, sizeof(LightCam));
    const Solid *sp = &sc->solids[sphere];
    const Light *l = &sc->lights[light];
    lc->sphere = sphere;
    lc->light = light;
    lc->light_pos = l->pos;
    lc->sphere_pos = sp->pos;
    double distToSolid = vnorm(vsub(sp->pos, l->pos));
    double angleOfView = atan2(sp->radius, distToSolid) * 2.82;
    Camera c;
    camera_setup(&c, l->pos, sp->pos, V(0, .5, 1), sc->bg, angleOfView, 1, 1, 1);
    int numSamples = sc->lightCamSamples;
    int cap = 0;
    for (int si = 0; si < numSamples; si++)
        for (int sj = 0; sj < numSamples; sj++) {
            lc->spw = c.pixWidth / numSamples;
            double p, q, r;
            if (numSamples > 1) {
                double edgeX = (0 - .5) * c.pixWidth;
                double edgeY = (0 - .5) * c.pixWidth;
                double j_m = edgeX + sj * lc->spw + .5 * lc->spw;
                double i_m = edgeY + si * lc->spw + .5 * lc->spw;
                p = c.l + j_m;
                q = c.t - i_m;
            } else {
                p = c.l + 0 * c.pixWidth;
                q = c.t - 0 * c.pixWidth;
            }
            r = -c.d;
            v3 rayDir = vadd(vadd(smul(p, c.u), smul(q, c.v)), smul(r, c.w));
            rayDir = vnormalized(rayDir);
            light_ray(sc, lc, &cap, rayDir, c.e, 20, 1.0, 1.0, 1);
        }
    double size = lc->maxDistFromZero * 2 + 1;
    lc->perm = (int *)malloc(sizeof(int) * (size_t)(lc->npts > 0 ? lc->npts : 1));
    for (int i = 0; i < lc->npts; i++) lc->perm[i] = i;
    int *scratch = (int *)malloc(sizeof(int) * (size_t)(lc->npts > 0 ? lc->npts : 1));
    unsigned char *which = (unsigned char *)malloc((size_t)(lc->npts > 0 ? lc->npts : 1));
    lc->root = build_pnode(lc, V(0 - size / 2, 0 - size / 2, 0 - size / 2), size, 0, lc->npts, scratch, which);
    free(scratch);
    free(which);
    return lc;
}

/* LightCam::getIntensity, LightCam.h:571-620 */
static double lc_get_intensity(const LightCam *lc, const PNode *r, double maxDist, v3 loc)
{
    if (!r) return 0;
    double total = 0;
    if (r->size < maxDist * 2 || r->leaf) {
        for (int k = r->begin; k < r->end; k++) {
            int id = lc->perm[k];
            double dist = vnorm(vsub(lc->pts[id], loc));
            if (dist < maxDist) total += lc->w[id];
        }
        return total;
    }
    for (int i = 0; i < 8; i++) {
        double epsilon = maxDist * 0.1;
        if (vnorm(vsub(loc, r->origin)) < (r->size + epsilon))             /* insideOctree2, LightCam.h:559-562 */
            total += lc_get_intensity(lc, r->child[i], maxDist, loc);
    }
    return total;
}

/* LightCam::calcIntensity, LightCam.h:121-203 */
static double lc_calc_intensity(const LightCam *lc, v3 pos)
{
    double smoothingFactor = 1.5;
    double dist = vnorm(vsub(pos, lc->light_pos));
    double sphereDistToLight = vnorm(vsub(lc->light_pos, lc->sphere_pos));
    double maxDist = lc->spw * (dist / sphereDistToLight);
    maxDist *= smoothingFactor;
    double total = lc_get_intensity(lc, lc->root, maxDist, pos);
    double expected = (maxDist * maxDist) / (lc->spw * lc->spw);
    return total / expected;
}

static LightCam *get_lightcam(Scene *sc, int sphere, int light)
{
    LightCam *found = NULL;
    pthread_mutex_lock(&sc->cam_lock);
    for (int i = 0; i < sc->ncams; i++)
        if (sc->cams[i]->sphere == sphere && sc->cams[i]->light == light) { found = sc->cams[i]; break; }
    if (!found) {
        found = lightcam_create(sc, sphere, light);
        if (sc->ncams == sc->ccams) {
            sc->ccams = sc->ccams ? sc->ccams * 2 : 8;
            sc->cams = (LightCam **)realloc(sc->cams, sizeof(LightCam *) * (size_t)sc->ccams);
        }
        sc->cams[sc->ncams++] = found;
    }
    pthread_mutex_unlock(&sc->cam_lock);
    return found;
}

/* ---------------------------------------------------------------- shading */
enum { ORC_FLAG_NO_LIGHTCAM = 1 };   /* treat "occluded by transparent spheres only" as shadowed */

typedef struct {
    Scene *sc;
    const Camera *cam;
    int flags;
    Counters ctr;
    unsigned char *lc_mask;    /* per pixel: a light-camera query happened while shading it */
    int cur_pixel;
} Ctx;

/* Camera::computeLocalColor Camera.h:365-488, Solid::computeDiffuseComp Solid.h:128-146,
 * Solid::computeSpecularComp Solid.h:151-172 */
static v3 compute_local_color(Ctx *cx, const HitRecord *hr)
{
    Scene *sc = cx->sc;
    const Solid *solid = &sc->solids[hr->nearest];
    v3 e = hr->e;
    v3 impactPos = hr->impactPos;
    v3 normal = vnormalized(hr->normal);                                   /* Camera.h:382 */
    v3 localColor = V(0, 0, 0);
    const double minT = .01;
    for (int li = 0; li < sc->nlights; li++) {
        const Light *l = &sc->lights[li];
        v3 rayDir = vnormalized(vsub(l->pos, impactPos));
        v3 distToLight = vsub(l->pos, impactPos);
        double k = .999999;
        int only = 1;
        int closest = -1;
        cx->ctr.shadow_rays++;
        int occluded = check_occluded3(sc, sc->root, hr->nearest, rayDir, impactPos, minT, vnorm(distToLight) - .01, &only,
                                       &closest, &cx->ctr.node_tests_shadow, &cx->ctr.prim_tests_shadow);
        if (occluded && only) {
            if (cx->lc_mask) cx->lc_mask[cx->cur_pixel] = 1;
            if (!(cx->flags & ORC_FLAG_NO_LIGHTCAM)) {
                LightCam *lc = get_lightcam(sc, closest, li);
                cx->ctr.lightcam_queries++;
                k = lc_calc_intensity(lc, impactPos);
                if (k > .01) occluded = 0;
            }
        }
        if (!occluded) {
            if (solid->m.Ks > 0) {
                v3 n = vnormalized(normal);
                v3 v = vnormalized(vsub(e, impactPos));
                v3 lDir = vnormalized(vsub(l->pos, impactPos));
                v3 reflectionDir = vnormalized(vadd(vneg(lDir), smul(2.0 * vdot(lDir, n), n)));
                double pw = pow(fmax(vdot(reflectionDir, v), 0.0), solid->m.Shine);
                v3 comp = vmul(smul(l->intensity * solid->m.Ks, l->color), pw);
                localColor = vadd(localColor, smul(k, comp));
            }
            double dm = fmax(vdot(vnormalized(vsub(l->pos, impactPos)), normal), 0.0);
            v3 comp = vmul(smul(l->intensity * solid->m.Kd, V(solid->m.r, solid->m.g, solid->m.b)), dm);
            localColor = vadd(localColor, smul(k, comp));
        }
    }
    return localColor;
}

/* Camera::castRay(rayDir, e, ...), Camera.h:318-363 */
static v3 cast_ray(Ctx *cx, v3 rayDir, v3 e, HitRecord *hr)
{
    const double maxT = 10000000000.0, minT = .01;
    hr->nearest = -1;
    hr->t = maxT;
    cx->ctr.closest_rays++;
    octree_intersect(cx->sc, cx->sc->root, rayDir, e, minT, maxT, hr, &cx->ctr.node_tests_closest, &cx->ctr.prim_tests_closest);
    if (hr->nearest < 0) return cx->cam->bg;
    return compute_local_color(cx, hr);
}

/* Camera::recursiveCastRays, Camera.h:232-314 */
static v3 recursive_cast_rays(Ctx *cx, v3 rayDir, v3 e, int maxReflections, double CoR)
{
    HitRecord hr[7];
    v3 color = cast_ray(cx, rayDir, e, &hr[0]);
    double coefOfReflection = CoR;
    for (int i = 0; i < maxReflections; i++) {
        if (hr[i].nearest < 0) break;
        if (coefOfReflection < .01) break;
        const Solid *s = &cx->sc->solids[hr[i].nearest];
        if (s->m.T > 0) {
            double ior = s->m.ior;
            v3 normal = hr[i].normal;
            v3 viewDir = vneg(hr[i].rayDir);
            double dot = vdot(normal, viewDir);
            int entering = (dot >= 0);
            if (entering) ior = 1.0 / ior;
            else { normal = vneg(normal); dot = -dot; }
            double discriminate = 1 - (1 - (dot * dot)) * (ior * ior);
            double T = s->m.T;
            if (discriminate > 0) {
                v3 refractionDir = vsub(vmul(vsub(vmul(normal, dot), viewDir), ior), vmul(normal, sqrt(discriminate)));
                color = vadd(color, smul(T, recursive_cast_rays(cx, refractionDir, hr[i].impactPos, maxReflections - i - 1,
                                                                 T * coefOfReflection)));
            } else {
                v3 ird = vnormalized(vadd(hr[i].rayDir, smul(2.0 * -vdot(hr[i].rayDir, normal), normal)));
                color = vadd(color, smul(T, recursive_cast_rays(cx, ird, hr[i].impactPos, maxReflections - i - 1,
                                                                 T * coefOfReflection)));
            }
        }
        coefOfReflection *= s->m.Ks;
        color = vadd(color, smul(coefOfReflection, cast_ray(cx, hr[i].reflectionDir, hr[i].impactPos, &hr[i + 1])));
    }
    return color;
}

/* Counter-based replacement for the reference's shared libc rand()
 * (Camera.h:99-103): value k/1000, k in [0,1000), keyed by pixel, sample and
 * dimension.  The reference's own stream is not reproducible across threads. */
static inline uint32_t orc_hash(uint32_t pixel, uint32_t sample, uint32_t dim, uint32_t seed)
{
    uint32_t h = seed * 0x9E3779B1u + 0x85EBCA6Bu;
    h ^= pixel + 0x9E3779B9u + (h << 6) + (h >> 2);
    h ^= sample + 0x9E3779B9u + (h << 6) + (h >> 2);
    h ^= dim + 0x9E3779B9u + (h << 6) + (h >> 2);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
static inline double normed_rand(uint32_t pixel, uint32_t sample, uint32_t dim, uint32_t seed)
{
    return (double)(orc_hash(pixel, sample, dim, seed) % 1000u) / 1000.0;
}

/* Camera::castRay(i, j, ...), Camera.h:112-230.  Also reports the first sample's primary hit. */
static v3 cast_pixel(Ctx *cx, int i, int j, uint32_t seed, int *hit_id, double *hit_t)
{
    const Camera *c = cx->cam;
    int DoF = c->aperture > 0;
    int maxReflections = 5;
    int N = c->numSamples;
    v3 overall = V(0, 0, 0);
    uint32_t pixel = (uint32_t)i * (uint32_t)c->resx + (uint32_t)j;
    for (int si = 0; si < N; si++)
        for (int sj = 0; sj < N; sj++) {
            uint32_t sample = (uint32_t)(si * N + sj);
            double spw = c->pixWidth / N;
            double p, q, r;
            if (N > 1) {
                double randX = normed_rand(pixel, sample, 0, seed);
                double randY = normed_rand(pixel, sample, 1, seed);
                randX -= .5;
                randY -= .5;
                double edgeX = (j - .5) * c->pixWidth;
                double edgeY = (i - .5) * c->pixWidth;
                double j_m = edgeX + sj * spw + .5 * spw;
                double i_m = edgeY + si * spw + .5 * spw;
                randX *= spw;
                randY *= spw;
                j_m += randX;
                i_m += randY;
                p = c->l + j_m;
                q = c->t - i_m;
            } else {
                p = c->l + j * c->pixWidth;
                q = c->t - i * c->pixWidth;
            }
            r = -c->d;
            v3 rayDir = vadd(vadd(smul(p, c->u), smul(q, c->v)), smul(r, c->w));
            rayDir = vnormalized(rayDir);
            v3 from = c->e;
            if (DoF) {
                v3 uOffset = vmul(smul(normed_rand(pixel, sample, 2, seed) - .5, c->u), c->aperture);
                v3 vOffset = vmul(smul(normed_rand(pixel, sample, 3, seed) - .5, c->v), c->aperture);
                v3 off = vadd(uOffset, vOffset);
                from = vadd(from, off);
                p = c->l + j * c->pixWidth;
                q = c->t - i * c->pixWidth;
                v3 focusPoint = vadd(vadd(vadd(smul(p, c->u), smul(q, c->v)), smul(r, c->w)), c->e);
                rayDir = vnormalized(vsub(focusPoint, from));
            }
            if (sample == 0 && hit_id) {
                /* primary hit of the first sample, for hit-ID parity maps (not counted) */
                HitRecord h0;
                uint64_t a = 0, b = 0;
                h0.nearest = -1;
                h0.t = 10000000000.0;
                octree_intersect(cx->sc, cx->sc->root, rayDir, from, .01, 10000000000.0, &h0, &a, &b);
                *hit_id = h0.nearest;
                if (hit_t) *hit_t = h0.t;
            }
            overall = vadd(overall, vdiv(recursive_cast_rays(cx, rayDir, from, maxReflections, 1.0), (double)(N * N)));
        }
    overall.x = fmax(0.0, fmin(1.0, overall.x));
    overall.y = fmax(0.0, fmin(1.0, overall.y));
    overall.z = fmax(0.0, fmin(1.0, overall.z));
    return overall;
}

/* ------------------------------------------------------------- public API */
typedef struct {
    Scene *sc;
    Camera cam;
    int flags, tid, nthreads, row0, row1;
    const int *rows;           /* optional explicit row list: then row0..row1 index into it */
    uint32_t seed;
    double *pixels;
    int *hit_ids;
    double *hit_t;
    unsigned char *lc_mask;
    Counters ctr;
} Job;

/* renderRows2, Driver.cpp:735-769: rows tid, tid+stride, ... ; value = colour * 255 */
static void *render_rows(void *arg)
{
    Job *jb = (Job *)arg;
    Ctx cx;
    memset(&cx, 0, sizeof cx);
    cx.sc = jb->sc;
    cx.cam = &jb->cam;
    cx.flags = jb->flags;
    cx.lc_mask = jb->lc_mask;
    int resx = jb->cam.resx;
    for (int k = jb->row0 + jb->tid; k < jb->row1; k += jb->nthreads) {
        const int i = jb->rows ? jb->rows[k] : k;
        for (int j = 0; j < resx; j++) {
            size_t px = (size_t)i * (size_t)resx + (size_t)j;
            cx.cur_pixel = (int)px;
            int hid = -1;
            double ht = 0;
            v3 c = cast_pixel(&cx, i, j, jb->seed, jb->hit_ids ? &hid : NULL, &ht);
            jb->pixels[px * 3 + 0] = c.x * 255;
            jb->pixels[px * 3 + 1] = c.y * 255;
            jb->pixels[px * 3 + 2] = c.z * 255;
            if (jb->hit_ids) jb->hit_ids[px] = hid;
            if (jb->hit_t) jb->hit_t[px] = ht;
        }
    }
    jb->ctr = cx.ctr;
    return NULL;
}

Scene *orc_load(const char *path, int phong)
{
    Scene *s = (Scene *)calloc(1, sizeof(Scene));
    s->lightCamSamples = 150;
    pthread_mutex_init(&s->cam_lock, NULL);
    if (!parse_nff(s, path)) { free(s); return NULL; }
    s->phong = phong;
    build_octree(s);
    return s;
}

void orc_free(Scene *s)
{
    if (!s) return;
    for (int i = 0; i < s->ncams; i++) {
        free(s->cams[i]->pts); free(s->cams[i]->w); free(s->cams[i]->perm); free_pnode(s->cams[i]->root); free(s->cams[i]);
    }
    free(s->cams);
    free_node(s->root);
    free(s->solids);
    free(s->lights);
    pthread_mutex_destroy(&s->cam_lock);
    free(s);
}

/* info[]: resx, resy, nsolids, nnodes, nlights, lightCamSamples, ntriangles, nspheres */
void orc_info(const Scene *s, int *info)
{
    int nt = 0, ns = 0;
    for (int i = 0; i < s->nsolids; i++) { if (s->solids[i].type == SOLID_TRI) nt++; else ns++; }
    info[0] = s->resx; info[1] = s->resy; info[2] = s->nsolids; info[3] = s->nnodes;
    info[4] = s->nlights; info[5] = s->lightCamSamples; info[6] = nt; info[7] = ns;
}

void orc_set_resolution(Scene *s, int resx, int resy) { s->resx = resx; s->resy = resy; }
void orc_set_lightcam_samples(Scene *s, int n) { s->lightCamSamples = n; }

/* Renders rows [row0,row1) of the frame into pixels[resy*resx*3] (value*255, unclamped
 * to integers) with `nthreads` row-interleaved threads.  counters[7] receives:
 * closest rays, shadow rays, node tests (closest), prim tests (closest), node tests
 * (shadow), prim tests (shadow), light-camera queries.  hit_ids / hit_t / lc_mask may be NULL. */
static int render_job(Scene *s, int numSamples, double aperture, int nthreads, int row0, int row1, const int *rows,
                      uint32_t seed, int flags, double *pixels, int *hit_ids, double *hit_t, unsigned char *lc_mask,
                      uint64_t *counters)
{
    Camera cam;
    memset(&cam, 0, sizeof cam);
    camera_setup(&cam, s->from, s->at, s->up, s->bg, s->angle, s->resx, s->resy, 0);
    cam.numSamples = numSamples;
    cam.aperture = aperture;
    Job *jobs = (Job *)calloc((size_t)nthreads, sizeof(Job));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int t = 0; t < nthreads; t++) {
        jobs[t].sc = s; jobs[t].cam = cam; jobs[t].flags = flags; jobs[t].tid = t; jobs[t].nthreads = nthreads;
        jobs[t].row0 = row0; jobs[t].row1 = row1; jobs[t].rows = rows; jobs[t].seed = seed; jobs[t].pixels = pixels;
        jobs[t].hit_ids = hit_ids; jobs[t].hit_t = hit_t; jobs[t].lc_mask = lc_mask;
        pthread_create(&th[t], NULL, render_rows, &jobs[t]);
    }
    Counters tot;
    memset(&tot, 0, sizeof tot);
    for (int t = 0; t < nthreads; t++) {
        pthread_join(th[t], NULL);
        tot.closest_rays += jobs[t].ctr.closest_rays;
        tot.shadow_rays += jobs[t].ctr.shadow_rays;
        tot.node_tests_closest += jobs[t].ctr.node_tests_closest;
        tot.prim_tests_closest += jobs[t].ctr.prim_tests_closest;
        tot.node_tests_shadow += jobs[t].ctr.node_tests_shadow;
        tot.prim_tests_shadow += jobs[t].ctr.prim_tests_shadow;
        tot.lightcam_queries += jobs[t].ctr.lightcam_queries;
    }
    if (counters) {
        counters[0] = tot.closest_rays; counters[1] = tot.shadow_rays; counters[2] = tot.node_tests_closest;
        counters[3] = tot.prim_tests_closest; counters[4] = tot.node_tests_shadow; counters[5] = tot.prim_tests_shadow;
        counters[6] = tot.lightcam_queries;
    }
    free(jobs);
    free(th);
    return 0;
}

int orc_render(Scene *s, int numSamples, double aperture, int nthreads, int row0, int row1, uint32_t seed, int flags,
               double *pixels, int *hit_ids, double *hit_t, unsigned char *lc_mask, uint64_t *counters)
{
    if (!s || s->resx <= 0 || s->resy <= 0 || nthreads < 1) return -1;
    if (row1 > s->resy) row1 = s->resy;
    if (row0 < 0) row0 = 0;
    return render_job(s, numSamples, aperture, nthreads, row0, row1, NULL, seed, flags, pixels, hit_ids, hit_t, lc_mask,
                      counters);
}

/* Same, for an explicit list of rows (threads interleave over the list), e.g. a sample of a large frame. */
int orc_render_rows(Scene *s, int numSamples, double aperture, int nthreads, const int *rows, int n_rows, uint32_t seed,
                    int flags, double *pixels, int *hit_ids, double *hit_t, unsigned char *lc_mask, uint64_t *counters)
{
    if (!s || s->resx <= 0 || s->resy <= 0 || nthreads < 1 || !rows || n_rows < 0) return -1;
    for (int k = 0; k < n_rows; k++)
        if (rows[k] < 0 || rows[k] >= s->resy) return -1;
    return render_job(s, numSamples, aperture, nthreads, 0, n_rows, rows, seed, flags, pixels, hit_ids, hit_t, lc_mask,
                      counters);
}

/* writeImage, Driver.cpp:712-733: (int) truncation, binary P6 */
int orc_write_ppm(const double *pixels, int resx, int resy, const char *path)
{
    size_t n = (size_t)resx * (size_t)resy * 3;
    unsigned char *b = (unsigned char *)malloc(n);
    for (size_t i = 0; i < n; i++) b[i] = (unsigned char)(int)pixels[i];
    FILE *f = fopen(path, "wb");
    if (!f) { free(b); return -1; }
    fprintf(f, "P6\n%d %d\n%d\n", resx, resy, 255);
    fwrite(b, 1, n, f);
    fclose(f);
    free(b);
    return 0;
}

void orc_to_u8(const double *pixels, size_t n, unsigned char *out)
{
    for (size_t i = 0; i < n; i++) out[i] = (unsigned char)(int)pixels[i];
}

/* ---- structure dumps, so that the product's host front end can be checked ---- */
/* per solid: type, 8 material doubles, 9 vertex doubles (or centre+radius), has_vn, 9 normals */
void orc_get_solid(const Scene *s, int id, int *type, double *mat8, double *geom9, int *has_vn, double *vn9)
{
    const Solid *x = &s->solids[id];
    *type = x->type;
    mat8[0] = x->m.r; mat8[1] = x->m.g; mat8[2] = x->m.b; mat8[3] = x->m.Kd;
    mat8[4] = x->m.Ks; mat8[5] = x->m.Shine; mat8[6] = x->m.T; mat8[7] = x->m.ior;
    memset(geom9, 0, sizeof(double) * 9);
    memset(vn9, 0, sizeof(double) * 9);
    *has_vn = x->has_vn;
    if (x->type == SOLID_TRI) {
        for (int k = 0; k < 3; k++) {
            geom9[k * 3] = x->v[k].x; geom9[k * 3 + 1] = x->v[k].y; geom9[k * 3 + 2] = x->v[k].z;
            vn9[k * 3] = x->vn[k].x; vn9[k * 3 + 1] = x->vn[k].y; vn9[k * 3 + 2] = x->vn[k].z;
        }
    } else {
        geom9[0] = x->pos.x; geom9[1] = x->pos.y; geom9[2] = x->pos.z; geom9[3] = x->radius;
    }
}

void orc_get_camera(const Scene *s, double *out17)
{
    Camera c;
    memset(&c, 0, sizeof c);
    camera_setup(&c, s->from, s->at, s->up, s->bg, s->angle, s->resx, s->resy, 0);
    double v[17] = {c.u.x, c.u.y, c.u.z, c.v.x, c.v.y, c.v.z, c.w.x, c.w.y, c.w.z, c.e.x, c.e.y, c.e.z,
                    c.d, c.pixWidth, c.l, c.t, c.distToEdge};
    memcpy(out17, v, sizeof v);
}

void orc_get_light(const Scene *s, int i, double *out7)
{
    const Light *l = &s->lights[i];
    out7[0] = l->pos.x; out7[1] = l->pos.y; out7[2] = l->pos.z;
    out7[3] = l->color.x; out7[4] = l->color.y; out7[5] = l->color.z; out7[6] = l->intensity;
}

void orc_get_background(const Scene *s, double *out3) { out3[0] = s->bg.x; out3[1] = s->bg.y; out3[2] = s->bg.z; }

/* depth-first pre-order walk (children 0..7): per node centre(3)+radius, solid count; solid ids concatenated */
static void dump_walk(const Node *n, double *node4, int *node_cnt, int *ids, int *ni, int *nid)
{
    if (node4) { node4[*ni * 4] = n->origin.x; node4[*ni * 4 + 1] = n->origin.y; node4[*ni * 4 + 2] = n->origin.z; node4[*ni * 4 + 3] = n->brad; }
    if (node_cnt) node_cnt[*ni] = n->nsolids;
    (*ni)++;
    for (int k = 0; k < n->nsolids; k++) { if (ids) ids[*nid] = n->solids[k]; (*nid)++; }
    for (int i = 0; i < 8; i++) if (n->child[i]) dump_walk(n->child[i], node4, node_cnt, ids, ni, nid);
}
/* returns total resident solid references; pass NULLs to size the arrays first */
int orc_dump_octree(const Scene *s, double *node4, int *node_cnt, int *ids)
{
    int ni = 0, nid = 0;
    dump_walk(s->root, node4, node_cnt, ids, &ni, &nid);
    return nid;
}

/* ---- single-primitive entry points for unit tests ---- */
/* returns t (or maxT); out7 = hit t, normal(3), reflectionDir(3) when hit */
double orc_test_triangle(const double *v9, const double *vn9, int phong, const double *d3, const double *e3, double minT,
                         double maxT, double *out7)
{
    Solid s;
    memset(&s, 0, sizeof s);
    s.type = SOLID_TRI;
    for (int k = 0; k < 3; k++) s.v[k] = V(v9[k * 3], v9[k * 3 + 1], v9[k * 3 + 2]);
    if (vn9) { s.has_vn = 1; for (int k = 0; k < 3; k++) s.vn[k] = V(vn9[k * 3], vn9[k * 3 + 1], vn9[k * 3 + 2]); }
    tri_finalize(&s);
    HitRecord hr;
    memset(&hr, 0, sizeof hr);
    hr.nearest = -1;
    hr.t = maxT;
    double t = tri_intersect(&s, 0, phong, V(d3[0], d3[1], d3[2]), V(e3[0], e3[1], e3[2]), minT, maxT, &hr);
    if (out7) { out7[0] = hr.t; out7[1] = hr.normal.x; out7[2] = hr.normal.y; out7[3] = hr.normal.z;
                out7[4] = hr.reflectionDir.x; out7[5] = hr.reflectionDir.y; out7[6] = hr.reflectionDir.z; }
    return t;
}

double orc_test_sphere(const double *c3, double radius, const double *d3, const double *e3, double minT, double maxT,
                       double *out7)
{
    Solid s;
    memset(&s, 0, sizeof s);
    s.type = SOLID_SPHERE;
    s.pos = V(c3[0], c3[1], c3[2]);
    s.radius = radius;
    HitRecord hr;
    memset(&hr, 0, sizeof hr);
    hr.nearest = -1;
    hr.t = maxT;
    double t = sphere_intersect(&s, 0, V(d3[0], d3[1], d3[2]), V(e3[0], e3[1], e3[2]), minT, maxT, &hr);
    if (out7) { out7[0] = hr.t; out7[1] = hr.normal.x; out7[2] = hr.normal.y; out7[3] = hr.normal.z;
                out7[4] = hr.reflectionDir.x; out7[5] = hr.reflectionDir.y; out7[6] = hr.reflectionDir.z; }
    return t;
}

double orc_test_bound(const double *c3, double radius, const double *d3, const double *e3, double minT, double maxT)
{
    return bound_intersect(V(c3[0], c3[1], c3[2]), radius, V(d3[0], d3[1], d3[2]), V(e3[0], e3[1], e3[2]), minT, maxT);
}

/* The same three tests for n items at once.  kind 0: triangle (geom: 9 doubles v0 v1 v2), 2: sphere (geom: centre,
 * radius), 3: node bounding sphere (geom: centre, radius).  rays: 6 doubles per item (origin, direction).
 * out: 4 doubles per item: the returned t, then the unit normal stored in the hit record (zero on a miss / kind 3). */
void orc_test_batch(int kind, int n, const double *geom, const double *rays, double minT, double maxT, double *out)
{
    for (int i = 0; i < n; i++) {
        const double *e = rays + (size_t)i * 6, *d = e + 3;
        double o7[7] = {0, 0, 0, 0, 0, 0, 0};
        double t;
        if (kind == 0) t = orc_test_triangle(geom + (size_t)i * 9, NULL, 0, d, e, minT, maxT, o7);
        else if (kind == 2) t = orc_test_sphere(geom + (size_t)i * 4, geom[(size_t)i * 4 + 3], d, e, minT, maxT, o7);
        else t = orc_test_bound(geom + (size_t)i * 4, geom[(size_t)i * 4 + 3], d, e, minT, maxT);
        out[(size_t)i * 4] = t;
        const int hit = kind != 3 && t < maxT;
        out[(size_t)i * 4 + 1] = hit ? o7[1] : 0;
        out[(size_t)i * 4 + 2] = hit ? o7[2] : 0;
        out[(size_t)i * 4 + 3] = hit ? o7[3] : 0;
    }
}

/* light-camera introspection: number of cameras built so far and their sizes */
int orc_lightcam_count(const Scene *s) { return s->ncams; }
void orc_lightcam_info(const Scene *s, int i, int *sphere, int *light, int *npts, uint64_t *rays)
{
    *sphere = s->cams[i]->sphere; *light = s->cams[i]->light; *npts = s->cams[i]->npts; *rays = s->cams[i]->light_rays;
}

#ifdef ORC_MAIN
#include <time.h>
static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
/* usage: rt_oracle scene.nff out.ppm [threads] [phong] [numSamples] */
int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s scene.nff out.ppm [threads] [phong] [numSamples]\n", argv[0]); return 2; }
    int threads = argc > 3 ? atoi(argv[3]) : 2;
    int phong = argc > 4 ? atoi(argv[4]) : 0;
    int ns = argc > 5 ? atoi(argv[5]) : 1;
    Scene *s = orc_load(argv[1], phong);
    if (!s) { fprintf(stderr, "error opening file\n"); return 1; }
    int info[8];
    orc_info(s, info);
    printf("res %dx%d solids %d nodes %d lights %d Z %d (tris %d spheres %d)\n", info[0], info[1], info[2], info[3], info[4],
           info[5], info[6], info[7]);
    double *px = (double *)calloc((size_t)info[0] * (size_t)info[1] * 3, sizeof(double));
    uint64_t c[7];
    double t0 = now_ms();
    orc_render(s, ns, 0.0, threads, 0, info[1], 0, 0, px, NULL, NULL, NULL, c);
    double t1 = now_ms();
    printf("render %.1f ms, closest %llu shadow %llu nodeC %llu primC %llu nodeS %llu primS %llu lcq %llu\n", t1 - t0,
           (unsigned long long)c[0], (unsigned long long)c[1], (unsigned long long)c[2], (unsigned long long)c[3],
           (unsigned long long)c[4], (unsigned long long)c[5], (unsigned long long)c[6]);
    orc_write_ppm(px, info[0], info[1], argv[2]);
    free(px);
    orc_free(s);
    return 0;
}
#endif
