// rtx_pack.h -- host-side packing of the reference's AoS scene (rtx_scene, include/rtx.h) into
// the records the gfx950 kernels read, plus the ray-independent precomputation the reference
// repeats per ray:
//   * inverseMatrix (raytracing.cpp:144-165) once per sphere instead of on every invtransform
//     call (:182), bugs included, in the reference's exact float expression order;
//   * the ray-independent head of intersecttriangle (:330-337, :355): edge vectors, unit normal,
//     plane offset, area.
// Everything is evaluated in IEEE binary32 with contraction off, so the GPU consumes constants
// bit-identical to what the reference recomputes per ray.
#ifndef RTX_PACK_H_
#define RTX_PACK_H_

#include <cstdint>
#include <vector>

#include "rtx.h"

namespace rtx {

// 80-byte hot record of one triangle: 5 x float4, read with wave-uniform (scalar) loads.
//   q0 = { n.x, n.y, n.z, d }            unit normal `cross` (:332,:335), d = -(cross . u) (:337)
//   q1 = { v.x, v.y, v.z, area }         vertex v, area = 0.5*sqrt(cross.cross) (:334)
//   q2 = { w.x, w.y, w.z, v1.x }         vertex w, v1 = u - v (:330)
//   q3 = { v1.y, v1.z, v2.x, v2.y }      v2 = w - v (:331)
//   q4 = { v2.z, e.x, e.y, e.z }         e = u - w = `vec3` (:355)
struct TriHot { float q[20]; };

// 16-byte hot record of one sphere: { c.x, c.y, c.z, r*r } ((s.radius)*(s.radius), :211)
struct SphHot { float cx, cy, cz, r2; };

// 16-byte filter record of one primitive (spheres first, then triangles, file order):
// conservative bounding sphere { c.x, c.y, c.z, -R^2 } (rtx_bounds.cpp); -R^2 = -inf: always passes.
struct BoundSphere { float cx, cy, cz, neg_r2; };

BoundSphere bound_sphere_plain(const rtx_sphere& s);
BoundSphere bound_sphere_affine(const rtx_sphere& s, const rtx_xmatrix& im, double guard);
BoundSphere bound_sphere_triangle(const rtx_triangle& t, double guard);

// Edge filter of a triangle (rtx_bounds.cpp): barycentric coordinates of the reference's own point P with a
// certified margin, run between the plane intersection and the three area square roots of the exact test.
struct TriEdge { float a1x, a1y, a1z, m, a2x, a2y, a2z, Q; };
TriEdge tri_edge_filter(const rtx_triangle& t);

// 64-byte filter record of one affine sphere for SHADOW rays (raytracing.cpp:571-576: the
// direction is inverse-transformed, the origin is not): rows of the inverse's linear part, the
// centre, -R^2 (inflated; -inf = always pass), the sphere's index, and for the shadow kernel's packet test R and ||A||_F, both rounded up.
struct AffineShadowRec { float ax, ay, az, bx, by, bz, cx, cy, cz, ox, oy, oz, neg_r2; int32_t index; float r_up, fro_up; };
AffineShadowRec affine_shadow_record(const rtx_sphere& s, const rtx_xmatrix& im, int index);

// Spatial clusters of up to 64 filter records each, for the two-level packet test (rtx_device.hip,
// closest_packet / shadow_pass_packet): a wave first tests its packet against the CLUSTER spheres (one
// cluster per lane), then against the 64 MEMBER spheres of every cluster that passed (one member per
// lane).  A cluster sphere encloses its members' (already inflated) bounding spheres, so a member that
// can pass some ray's filter implies a cluster that passes the packet test.  Members that cannot be
// bounded (-R^2 = -inf or NaN: always pass) sit in clusters of their own whose record is -inf as well.
// Closest-hit needs the primitives in FILE order (strict `<`, sticky flag), so its clusters are built
// per segment of kSegPrims consecutive primitives and only mark bits in a per-wave bitmap that is then
// walked in index order; shadow rays are any-hit and test members as they come.
constexpr int kClusterSize = 64;
constexpr int kSuperRun = 8;     // divides closest_pool's kPoolQ (8 or 32; static_assert there)
constexpr int kSegPrims = 1 << 14;
struct ClusterSet {
    std::vector<BoundSphere> cl;        // [ncl]
    std::vector<BoundSphere> mem;       // [ncl * 64], padding: idx = -1
    std::vector<int32_t>     idx;       // [ncl * 64] primitive index (sphere i | ns + triangle i)
    std::vector<int32_t>     seg_first; // [nseg + 1] cluster range of every segment
    // per-ray traversal order (closest_lanes): clusters of affine spheres first (they never share a cluster
    // with anything else), then the rest; nfirst = how many come first
    std::vector<int32_t>     order;
    int32_t                  nfirst = 0;
    // the same hierarchy for the per-ray cone test (rtx_device.hip, cone_test): { c, R } with R >= 0 in
    // the 4th component (+inf: always passes); clR [ncl] = the cluster spheres, subR [ncl * 8] = spheres
    // around members 8g .. 8g+7 of each cluster
    std::vector<BoundSphere> clR, subR;
    // supR: { c, R } around every run of kSuperRun consecutive entries of `order`, the affine-first part and the rest
    // counted apart (closest_pool, stage A takes one or four runs a round): a wave none of whose rays passes a round's
    // spheres skips the round.  Consecutive clusters are neighbours in k-d order, so the runs are compact.
    std::vector<BoundSphere> supR;
};
// cls (optional, per record): records with a non-zero class are clustered apart from the others
void build_clusters(const std::vector<BoundSphere>& rec, const std::vector<int32_t>& ids, int seg_prims,
                    const std::vector<uint8_t>* cls, ClusterSet& out);

struct PackedScene {
    ClusterSet cc;                            // closest-hit clusters over `bounds`, segmented
    ClusterSet sc;                            // shadow clusters over `sbounds`, one segment
    // The per-ray traversal (k_closest_lanes) keeps the best (t, index) pair and needs no file order: scenes of more than
    // one segment get a set of their own for it, built over ALL records at once (seven overlapping hierarchies over 100,000
    // triangles otherwise, each cluster seven times the volume).  Empty when cc has one segment (cc serves both).
    ClusterSet pc;
    std::vector<AffineShadowRec> aff_shadow;  // one per affine sphere, file order
    // Shadow rays are any-hit: the order primitives are tried in cannot change the result, only how
    // soon a blocked ray stops.  sbounds = filter records of the shadow casters (spheres and triangles; polygons
    // cast none, :569-589) except affine spheres (aff_shadow: the reference transforms only the direction there,
    // :571-576), largest expected occluder first; sorder[i] = primitive index (sphere i | ns + triangle i).
    std::vector<BoundSphere> sbounds;
    std::vector<int32_t>     sorder;
    std::vector<BoundSphere> bounds;     // [nspheres + ntriangles], closest-hit call pattern
    float guard = 1.0f;                  // rays with an origin component beyond this bypass the filter
    std::vector<SphHot>      sph_hot;
    std::vector<int32_t>     sph_xf;     // istransformed (:34)
    std::vector<rtx_xmatrix> sph_im;     // inverseMatrix(mat) for EVERY sphere (needed for the stale
                                         // `transformed` flag quirk, :396-399, :443, :454)
    std::vector<rtx_sphere>  sph;        // raw records (materials, centre, radius, mat)
    std::vector<TriHot>      tri_hot;
    std::vector<TriEdge>     tri_edge;   // [ntriangles]
    // one 80-byte record per primitive (spheres first), fetched by the lanes that need it with five
    // independent 16-byte loads: sphere { c, r*r | im.a, im.b.x | im.b.yz, im.c.xy | im.c.z, im.d | xf, 0, 0, 0 },
    // triangle = its TriHot
    std::vector<TriHot>      prim_hot;
    // materials packed for the set-up and shading kernels (round 3: they gathered a dozen scalars out of the 120- / 88- / 84-byte
    // AoS records per hit -- one cache lookup per lane and scalar), four floats a row, copies of the records' own binary32 fields:
    //   mat4[k * 4 + 0] = { reflectivity, transparency, refractive_index, radius (spheres; 0 otherwise) }
    //   mat4[k * 4 + 1] = { color.xyz, specexpo }   [+ 2] = { speccoeff.xyz, istransformed }   [+ 3] = { diffcolor.xyz, 0 }
    std::vector<float>       mat4;       // [(nspheres + ntriangles + npolygons (+ planes)) * 16]
    std::vector<rtx_triangle> tri;
    std::vector<rtx_light>   lights;
    // polygons (raytracing.cpp:71-80, :289-325): raw records for the materials, vertices as float4 (x y z 0);
    // the ray-independent part of intersectpolygon -- the negated normal, d = -(n . v0) (:309-312) and the area
    // of the centroid fan (:320) -- sits in prim_hot: { n.xyz, d | totalArea, first_vertex, numvertices, 0 | 0.. }
    std::vector<rtx_polygon> poly;
    std::vector<float>       poly_verts;  // 4 floats per vertex
    // camera (raytracing.cpp:725-737)
    rtx_vec3 ncrossv, up, vpn, camera, orig, ambient;
    std::vector<float> imagex;           // per column, :753 evaluated in double, stored as float
    std::vector<float> imagey;           // per row,    :754
    uint64_t singular = 0;               // spheres with istransformed and det == 0 (:148)
    int max_fanout = 0;                  // max children one hit can spawn: [refl>0] + [transp>0]
};

int validate_scene(const rtx_scene* sc);  // RTX_OK or RTX_E_INVALID_ARG: pointers, counts, polygon vertex ranges
rtx_xmatrix inverse_matrix(const rtx_xmatrix& m, bool* singular);
// flags: RTX_F_TRACE_PLANES appends the planes behind the polygons, RTX_F_TRI_MOLLER_TRUMBORE packs the triangles' hot
// records for triangle_test_mt
void pack_scene(const rtx_scene& sc, int width, int height, PackedScene& out, uint32_t flags);

}  // namespace rtx
#endif
