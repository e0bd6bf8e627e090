// rtx_cells.hip.h -- shadow rays through a per-(light, direction cell) candidate table: k_cells_build (once per context), k_shadow_cells (every generation)
// Part of the one translation unit rtx_device.hip (device code, namespace rtx); not a stand-alone header.
#ifndef RTX_CELLS_HIP_H_
#define RTX_CELLS_HIP_H_

namespace rtx {

// ------------------------------------------------------------------------------------------------
// Every shadow ray's line passes through its light (raytracing.cpp:561-568: the direction is normalize(light - hit
// point), the origin the hit point moved 1e-4 along the normal), so WHICH plain primitives a shadow ray can touch is a
// function of its light, of the direction it arrives from and of how far from the light it starts -- nothing else.
// k_shadow_pass rediscovers that set per wave and frame: a packet per 64 rays, a walk of the cluster hierarchy, two
// dozen survivors broadcast one at a time.  Here it is tabulated ONCE per context:
//
//   cell_off[light * G * G + cell]   CSR offsets, cell = light_cell(direction, G) (octahedral map, Morton order)
//   cell_ent[...] = { primitive, dneed }   the shadow casters (plain spheres, triangles) some ray of that cell can
//                                          pass, sorted by dneed: a ray whose origin is nearer than dneed to the light
//                                          cannot pass this entry nor any later one
//
// and a shadow ray of ANY generation walks its own cell's list (one lane = one ray, four entries a step, stopping at
// the first dneed beyond its own distance), runs the per-ray filter on those, and hands the surviving (ray, primitive)
// pairs to the pooled exact stages of shadow_plain_pool (S / T / A: 64 pairs at a time, whoever they belong to).  No
// packet is built, no hierarchy walked, no survivor broadcast; incoherent rays (generations >= 1) cost what coherent
// ones do.
//
// Conservative by the packet argument that k_shadow_pass rests on, evaluated with the CELL as the packet
// (as k_affine_cells does for affine spheres):
//   * directions: the unit directions that light_cell maps to one cell are decode(q) / |decode(q)| for q in the cell's
//     square; decode is linear on each octahedron face, the square lies in one face or is cut by a face diagonal
//     through two of its corners, and the angle to a fixed axis over a planar convex piece is largest at a vertex: every
//     direction of the cell is within the largest of the four corner angles of the centre direction (sine taken from
//     the cross product, 1e-5 added for the rounding of light_cell at the borders).
//   * origins: apex-at-light packets (make_packet, apex_ok): rho = 1.3e-4 + 3.8e-3 D covers the origin bias, the
//     filter's angular tolerance and the rounding of the direction for every ray that starts within D of the light.
//     packet_test is monotone in rho (every operation on the path from rho to the comparison is a rounded monotone
//     one), so the SMALLEST of 33 levels D_j = dmax j / 32 at which an entry passes is a bound: rays with
//     D <= D_(j-1) cannot pass it.
//   * front: a primitive whose bounding sphere lies behind the origin of every ray of the cell that starts within D
//     of the light, with filt_front's margin (n.(o - c) >= 1.005 R + 1e-3 |o - c|), cannot be hit (t >= 0 only).
// dneed = max of the two.  Rays the table does not cover -- abnormal ones, origins further than dmax from the light or
// on top of it, zero directions -- take the transposed scan over every shadow caster, as in k_shadow_pass.
// ------------------------------------------------------------------------------------------------
constexpr int kCellLevels = 32;
constexpr int kCellCap = 512;    // entries one build wave sorts in LDS; longer lists are stored unsorted with dneed = -1
constexpr int kCellPad = 4;      // readable entries behind the last list (the walk loads four at a time)

__host__ __device__ __forceinline__ F3 oct_decode(float qx, float qy)
{
    const float z = 1.0f - fabsf(qx) - fabsf(qy);
    if (z >= 0.0f) return f3(qx, qy, z);
    return f3((1.0f - fabsf(qy)) * (qx >= 0.0f ? 1.0f : -1.0f), (1.0f - fabsf(qx)) * (qy >= 0.0f ? 1.0f : -1.0f), z);
}

// axis, sin(half-angle) rounded up, cos(half-angle) rounded down of the cone around cell m of the G x G map
struct CellCone { F3 w; float sT, cLo; };

__host__ __device__ __forceinline__ CellCone cell_cone(unsigned m, int G)
{
    const float gx = (float)demorton12(m), gy = (float)demorton12(m >> 1), s = 2.0f / (float)G;
    const F3 vc = oct_decode((gx + 0.5f) * s - 1.0f, (gy + 0.5f) * s - 1.0f);
    const float in = 1.0f / sqrtf(vc.x * vc.x + vc.y * vc.y + vc.z * vc.z);
    CellCone K;
    K.w = f3(vc.x * in, vc.y * in, vc.z * in);
    float sm = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const F3 v = oct_decode((gx + (float)(u & 1)) * s - 1.0f, (gy + (float)(u >> 1)) * s - 1.0f);
        const float cx = K.w.y * v.z - K.w.z * v.y, cy = K.w.z * v.x - K.w.x * v.z, cz = K.w.x * v.y - K.w.y * v.x;
        sm = fmaxf(sm, sqrtf(cx * cx + cy * cy + cz * cz) / sqrtf(v.x * v.x + v.y * v.y + v.z * v.z));
    }
    K.sT = sm * 1.0001f + 1e-5f;
    K.cLo = sqrtf(fmaxf(0.0f, 1.0f - K.sT * K.sT)) - 1e-6f;
    return K;
}

// the packet of make_packet(apex_ok) for a cone given directly; rho is set by the caller
__device__ __forceinline__ Packet cell_packet(const CellCone& K, F3 apex)
{
    Packet P;
    P.co = apex; P.w = K.w; P.rho = 0.0f; P.c0 = K.cLo; P.ok = true;
    const float cosE = 0.99999316f, sinE = 3.7e-3f;  // the filter's own angular tolerance, as make_packet
    const float cT = K.cLo * cosE - K.sT * sinE - 4e-6f;
    const float sT2 = K.sT * cosE + fminf(K.cLo + 2e-6f, 1.0f) * sinE + 4e-6f;
    P.cosT2 = cT * cT;
    P.sinT = sT2;
    return P;
}

__device__ __forceinline__ float cell_rho(float D) { return 1.3e-4f + 3.8e-3f * D; }  // make_packet's, for one ray at distance D

// ------------------------------------------------------------------------------------------------
// TIGHT entries (the default).  The per-ray filter's pass region carries the FILTER's own slack -- a cone of 3.7e-3 rad
// about the ray's origin for its binary32 evaluation -- and a table built on it lists what the filter can pass.  What a
// list needs to hold is less: the primitives whose EXACT test can accept some ray of the cell.  rtx_bounds.cpp states
// what acceptance implies about the ray's TRUE line (origin o, unit direction n, as real numbers):
//   * triangle: the plane point P the test computes (raytracing.cpp:352) lies within eta = 8u (guard + |c| + R) of the
//     line and inside the ball (c, R) of the filter record (R = Rs + eta: bound_sphere_triangle), and -- for triangles
//     with an edge record, Q > 0 -- has every barycentric coordinate beta_i(P) >= -m_i as the kernel evaluates them
//     (tri_edge_filter; m_0 = 2m), hence >= -1.5 m_i as real numbers (the evaluation error is < m / 4);
//   * plain sphere: the discriminant the test computes (:205-212) differs from the real one by <= 12u (2 |X|^2 + r^2) in the
//     filter's units, X = o - c, so dist(c, line)^2 <= r^2 (1 + 2^-14) + 64u |X|^2 -- the budget the filter itself grants
//     (the 2^-18 direction scale) -- i.e. dist <= R + 2.0e-3 |X| with R the filter record's radius.
// and every ray's line passes within delta = 2e-4 + 1e-6 dmax of its light (origin bias 1e-4 |normal|, rounding of the
// normalised direction).  So, with the cell's cone (axis w, half-angle theta, NO widening) and rho the slack named:
//   ball:      packet_test(cone, rho) on the filter record; triangles rho = delta; spheres rho = delta + 2.0e-3 X_j with
//              X_j >= |o - c| for every origin within D_j of the light that the front test keeps: levels as before;
//   triangle:  the plane through the light and each edge line of the grown triangle (tri_cell_reject below): a cell whose
//              cone lies wholly on the outer side of one of them holds no ray the test can accept.  NaN anywhere keeps the entry.
// The front test (dfront) is unchanged.  The per-ray filter stays in the walk as a cheap, sound first test.
// ------------------------------------------------------------------------------------------------
constexpr float kExactSphereTan = 2.0e-3f;  // sqrt(64u / (1 - 64u)) = 1.953e-3, rounded up

// true: NO ray of the cell can be accepted by triangle k's exact test.  q0 = { unit normal N, d }, q1 = { v, area } of the hot
// record, e0 = { a1, m }, e1 = { a2, Q } of the edge record, mb the filter record (the ball).  Edge i of the grown triangle is the
// line l_i = { X in the triangle's plane : beta_i(X) = -m_i' }, beta_i(X) = k_i + a_i.(X - v); the plane through the light and l_i
// (normal n_i, turned towards a_i) cuts the triangle's plane into beta_i >= -m_i' and the rest, so a line through the light with
// direction u meets the triangle's plane on the inside of edge i iff sign(sigma) n_i.u >= 0, sigma the axial coordinate of the
// meeting point (its sign is known when the ball does not straddle the light).  A line that misses the light by |e| <= delta
// moves that by |e| / |sigma|.  The accepted point P is not the meeting point X* itself: it lies within eta of the line and
// within ~eta of the plane, hence within 40 eta of X* unless the ray grazes the plane (|N.u| < 0.05: no decision then), which
// m_i' = 1.5 m_i + 40 eta |a_i| + 1e-4 absorbs.
__device__ __forceinline__ bool tri_cell_reject(const CellCone& K, const F3 Lp, const float4 q0, const float4 q1, const float4 e0, const float4 e1,
                                                const float4 mb, const float delta, const float eta)
{
    if (!(e1.w > 0.0f)) return false;  // no edge record (degenerate, huge, non-finite): the ball decides
    const F3 u0 = f3(-K.w.x, -K.w.y, -K.w.z);
    const float Nu = q0.x * u0.x + q0.y * u0.y + q0.z * u0.z;
    if (!(fabsf(Nu) - K.sT > 0.05f)) return false;  // some ray of the cell may graze the plane
    const float Cx = mb.x - Lp.x, Cy = mb.y - Lp.y, Cz = mb.z - Lp.z;
    const float C2 = Cx * Cx + Cy * Cy + Cz * Cz, Cn = sqrtf(C2);
    const float Cu = Cx * u0.x + Cy * u0.y + Cz * u0.z;
    const float hC = sqrtf(fmaxf(0.0f, C2 - Cu * Cu)) + 1e-3f * Cn;
    const float Rb = sqrtf(-mb.w) * 1.00001f;
    // axial range (along u = -n, from the light) of the points of a cell's line that lie within eta of the ball
    const float slo = fminf(Cu * K.cLo, Cu) - hC * K.sT - Rb - delta - eta - 1e-4f * Cn;
    const float shi = fmaxf(Cu * K.cLo, Cu) + hC * K.sT + Rb + delta + eta + 1e-4f * Cn;
    if (!(slo > 0.0f) && !(shi < 0.0f)) return false;  // the ball straddles the light
    const float off = (delta + eta) / (slo > 0.0f ? slo : -shi) + 6e-5f;  // what missing the light by delta + eta can turn a line by, + the rounding of n_i (<= 2.4e-5 under the guards below)
    const float m = e0.w;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float ax = i == 0 ? -(e0.x + e1.x) : (i == 1 ? e0.x : e1.x);
        const float ay = i == 0 ? -(e0.y + e1.y) : (i == 1 ? e0.y : e1.y);
        const float az = i == 0 ? -(e0.z + e1.z) : (i == 1 ? e0.z : e1.z);
        const float ki = i == 0 ? 1.0f : 0.0f;
        const float an2 = ax * ax + ay * ay + az * az, an = sqrtf(an2);
        const float mi = (i == 0 ? 3.0f : 1.5f) * m + 40.0f * eta * an + 1e-4f;
        // a point of l_i and its direction (in the plane, along which beta_i is constant)
        const float lam = (-mi - ki) / an2;
        const float wx = q1.x + lam * ax - Lp.x, wy = q1.y + lam * ay - Lp.y, wz = q1.z + lam * az - Lp.z;  // p_i - light
        const float tx = q0.y * az - q0.z * ay, ty = q0.z * ax - q0.x * az, tz = q0.x * ay - q0.y * ax;      // N x a_i
        float nx = ty * wz - tz * wy, ny = tz * wx - tx * wz, nz = tx * wy - ty * wx;                        // t_i x (p_i - light)
        const float nn = sqrtf(nx * nx + ny * ny + nz * nz);
        const float tn = sqrtf(tx * tx + ty * ty + tz * tz), wn = sqrtf(wx * wx + wy * wy + wz * wz);
        if (!(nn > 1e-2f * tn * wn)) continue;  // the light lies (nearly) on l_i itself: no decision from this edge
        const float na = nx * ax + ny * ay + nz * az;
        if (!(fabsf(na) > 1e-2f * nn * an)) continue;  // the light lies (nearly) in the triangle's plane: which side is inside is not safe to say
        const float sgn = na >= 0.0f ? 1.0f / nn : -1.0f / nn;
        nx *= sgn; ny *= sgn; nz *= sgn;
        const float g0 = nx * u0.x + ny * u0.y + nz * u0.z;
        const float gp = sqrtf(fmaxf(0.0f, 1.0f - g0 * g0)) + 1e-5f;
        const float glo = fminf(g0 * K.cLo, g0) - gp * K.sT, ghi = fmaxf(g0 * K.cLo, g0) + gp * K.sT;
        if (slo > 0.0f) { if (ghi < -off) return true; }   // every line meets the plane outside edge i
        else { if (glo > off) return true; }
    }
    return false;
}

// One wave per (light, cell), ONE pass: collect the cell's entries in LDS, take a range of `ent` with one atomic add on the table's
// cursor, sort by dneed, store; oc[cell] = { first entry, entries }.  Lists lie in the order the waves finish, which is why a list
// carries its own length.  A table that does not fit `cap` entries leaves oc = { 0, 0 } for the cells that came too late and the cursor
// at what the whole table needs: the host builds it again at that size (or takes a coarser map) -- a count pass in front, as the
// first version had, cost as much as the fill (C4: 2.2 of 6.4 ms; context creation is inside a scene's first rtx_render_rgb8 call).
// (Sixteen cells a workgroup, one atomic for the lot: atomics on one address are served one at a time, ~12 ns each, and a million of
// them -- one per cell of C4's map -- took 12 ms where the build itself takes 3.)
constexpr int kCellsPerGroup = 16;
// CAMERA mode (cam.on; round 3): the same table for the PRIMARY rays, which all leave one point as the shadow rays of a light all pass
// through one -- apex = the camera, one "light", cells keyed by the ray's own direction, entries = every primitive (the closest-hit
// cluster set) whose exact test can accept some ray of the cell:
//   plain sphere   dist(c, line) <= R + 2.0e-3 |X| (above) with X = apex - c for every ray: rho = delta + 2.0e-3 |C|
//   triangle       the ball with rho = delta, then the edge planes through the apex (tri_cell_reject), as for a light
//   affine sphere, polygon, plane   by what the per-ray FILTER can pass (cell_packet: the cone widened by the filter's own
//                  tolerance about the rays' common origin, rho = cell_rho(0), as k_closest_f's packet of a wave of primary rays)
//   front          plain spheres and triangles wholly behind the apex for every direction of the cell are left out (t >= 0 only)
// dneed carries the ORDER instead: -2 for affine spheres, -1 for the rest -- a ray tests its cell's affine spheres first (LexAcc).
// A list longer than kCellCap, or a cell outside the view cone (cam.axis, cam.cosv: the host's bound on the frame's ray directions),
// is marked { 0, -1 }: its rays take the scan over every primitive.
struct CamCells { int on; F3 apex; F3 axis; float cosv, sinv; };

__global__ __launch_bounds__(64 * kCellsPerGroup) void k_cells_build(SceneDev S, int G, const float* dmax, int2* oc, int2* ent, unsigned long long* cursor,
                                                                     unsigned long long cap, int tight, CamCells cam)
{
    __shared__ int2 bufs[kCellsPerGroup][kCellCap];
    __shared__ int s_n[kCellsPerGroup];
    __shared__ unsigned long long s_base;
    const int wv = threadIdx.x >> 6;
    int2* const buf = bufs[wv];
    const int cell = blockIdx.x * kCellsPerGroup + wv, lane = threadIdx.x & 63;
    const int li = cell / (G * G), m = cell % (G * G);
    const CellCone K = cell_cone((unsigned)m, G);
    const F3 Lp = cam.on ? cam.apex : f3(S.lights[li].point);
    Packet P = cell_packet(K, Lp);
    const float dm = dmax[li];
    P.rho = cam.on ? cell_rho(0.0f) : cell_rho(dm);
    // camera: a cell no ray of the frame maps to is not built (angle(axis, cell axis) > view half-angle + cell half-angle)
    const bool outside = cam.on && (K.w.x * cam.axis.x + K.w.y * cam.axis.y + K.w.z * cam.axis.z) < cam.cosv * K.cLo - cam.sinv * K.sT - 1e-4f;
    const float4* const clu = cam.on ? S.ccl : S.scl;
    const float4* const mem = cam.on ? S.cmem : S.smem;
    const int* const midx = cam.on ? S.cidx : S.sidx;
    const int nclu = outside ? 0 : (cam.on ? S.cseg[S.cnseg] : S.sncl);
    // tight entries: the cone as it is (no widening by the filter's tolerance), rho per entry
    const float delta = 2e-4f + 1e-6f * dm;
    Packet P0 = P;
    {
        const float c0 = fmaxf(0.0f, K.cLo - 4e-6f);
        P0.cosT2 = c0 * c0;
        P0.sinT = K.sT + 4e-6f;
        P0.rho = delta + kExactSphereTan * (3.3f * dm) * 1.01f;  // >= every entry's own rho (rho_of below: D - s_lo + hX <= dmax + 2.1 |C|, |C| <= dmax)
    }
    const Packet& Pc = tight ? P0 : P;  // what the cluster spheres are tested with
    int base = 0;
    int n = 0;
    // pass 0 collects (the first kCellCap entries in LDS); pass 1 -- only for a cell of more entries than that -- stores the rest
    for (int pass = 0; pass < 2; pass++) {
    n = 0;
    for (int cb = 0; cb < nclu; cb += 64) {
        const int c = cb + lane;
        // (camera: the filter-conservative packet P for the cluster test -- it contains the tight one's pass region as long as
        // rho(P0) <= what P's widening gives, which holds for every bounded member: 2.0e-3 |C| < 3.7e-3 |C|; affine spheres need P)
        unsigned long long cm = __ballot(c < nclu && (packet_test(Pc, clu[min(c, nclu - 1)]) || (cam.on && packet_test(P, clu[min(c, nclu - 1)]))));
        while (cm) {
            const size_t mi = (size_t)(cb + (__ffsll((long long)cm) - 1)) * 64 + lane;
            cm &= cm - 1;
            const float4 mb = mem[mi];
            const int k = midx[mi];
            bool keep = false;
            float dlev = -1.0f;
            // C = c - light; s = axial coordinate of the centre on the origins' side, h = its distance from the axis
            const float Cx = mb.x - Lp.x, Cy = mb.y - Lp.y, Cz = mb.z - Lp.z;
            const float C2 = Cx * Cx + Cy * Cy + Cz * Cz, Cn = sqrtf(C2);
            const float wC = K.w.x * Cx + K.w.y * Cy + K.w.z * Cz;
            const float h = sqrtf(fmaxf(0.0f, C2 - wC * wC)) + 1e-3f * Cn;
            const float R = sqrtf(-mb.w) * 1.00001f;
            const bool loose = !tight || (cam.on && (k >= S.ns + S.nt || (k < S.ns && S.sph_xf[max(k, 0)] != 0)));
            if (k >= 0 && loose) {
                keep = packet_test(P, mb);
                if (keep && !cam.on) {
                    // smallest level at which the entry passes (it passes at the last one)
                    int lo = 0, hi = kCellLevels;
                    Packet Pm = P;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        Pm.rho = cell_rho(dm * ((float)mid * (1.0f / (float)kCellLevels)));
                        if (packet_test(Pm, mb)) hi = mid; else lo = mid + 1;
                    }
                    dlev = (lo == 0) ? -1.0f : dm * ((float)(lo - 1) * (1.0f / (float)kCellLevels)) * 0.999999f;
                }
            } else if (k >= 0 && k >= S.ns) {
                // triangle: the ball (c, R) with the true line, then the edge half-spaces
                Packet Pm = P0;
                Pm.rho = delta;
                keep = packet_test(Pm, mb);
                if (keep && !S.mt) {  // (the Moeller-Trumbore mode has no edge records: the ball decides)
                    const float eta = (S.guard + Cn + sqrtf(Lp.x * Lp.x + Lp.y * Lp.y + Lp.z * Lp.z) + R) * (8.0f / 16777216.0f) * 1.01f;
                    const float4 q0 = S.tri_hot[(size_t)(k - S.ns) * 5], q1 = S.tri_hot[(size_t)(k - S.ns) * 5 + 1];
                    const float4 e0 = S.tri_edge[(size_t)(k - S.ns) * 2], e1 = S.tri_edge[(size_t)(k - S.ns) * 2 + 1];
                    if (tri_cell_reject(K, Lp, q0, q1, e0, e1, mb, delta, eta)) keep = false;
                }
            } else if (k >= 0) {
                // plain sphere: rho grows with the distance |X| of the origin from the centre; for origins within D_j of the
                // light that the front test keeps (D >= s - 1.2 R - ...), |X| <= max(D_j - s_lo, 1.2 R + ...) + h
                const float s_lo = -wC - h * K.sT - fabsf(wC) * (1.0f - K.cLo) - 1e-3f * Cn;
                const float xmin = 1.2f * R + 2e-3f * (Cn + dm) + 1e-2f;
                const float hX = h + fabsf(wC) * K.sT;  // distance of the centre from a LINE of the cell (the axis turned by theta)
                auto rho_of = [&](float D) { return delta + kExactSphereTan * ((fmaxf(D - s_lo, xmin) + hX) * 1.001f); };
                Packet Pm = P0;
                Pm.rho = cam.on ? delta + kExactSphereTan * (Cn * 1.001f + 1e-3f) : rho_of(dm);
                keep = packet_test(Pm, mb);
                if (keep && !cam.on) {
                    int lo = 0, hi = kCellLevels;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        Pm.rho = rho_of(dm * ((float)mid * (1.0f / (float)kCellLevels)));
                        if (packet_test(Pm, mb)) hi = mid; else lo = mid + 1;
                    }
                    dlev = (lo == 0) ? -1.0f : dm * ((float)(lo - 1) * (1.0f / (float)kCellLevels)) * 0.999999f;
                }
            }
            // behind every origin within D of the light: n.C <= ub for every direction n of the cell
            const float ub = (wC < 0.0f ? wC * K.cLo : wC) + h * K.sT;
            // (camera: the rays START at the apex and n is their direction -- a plain sphere or triangle with n.C + R < 0 for every n is behind them all)
            if (cam.on && keep && !loose && ub + 1.005f * R + 1e-3f * Cn + 1e-3f < 0.0f) keep = false;
            const unsigned long long bal = __ballot(keep);
            if (keep) {
                const float dfront = -ub - 1.005f * R - 1e-3f * (Cn + dm) - 1e-3f;
                float dneed = fmaxf(dlev, dfront);
                if (!(R < 1e18f) || dneed != dneed) dneed = -1.0f;  // records that always pass (R = inf): every ray takes them
                if (cam.on) dneed = (k < S.ns && S.sph_xf[k] != 0) ? -2.0f : -1.0f;
                const int pos = n + lanes_below(bal);
                if (pass == 0) { if (pos < kCellCap) buf[pos] = make_int2(k, __float_as_int(dneed)); }
                else if (pos >= kCellCap) ent[(size_t)base + pos] = make_int2(k, __float_as_int(-1.0f));
            }
            n += __popcll(bal);
        }
    }
    if (pass == 1) return;  // (the first kCellCap entries were stored after pass 0)
    {
        const bool nolist = cam.on && (outside || n > kCellCap);   // camera: its rays take the scan over every primitive
        if (lane == 0) s_n[wv] = nolist ? 0 : n;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int j = 0; j < kCellsPerGroup; j++) { before += (j < wv) ? s_n[j] : 0; total += s_n[j]; }
        if (threadIdx.x == 0) s_base = atomicAdd(cursor, (unsigned long long)total);
        __syncthreads();
        const unsigned long long b64 = s_base + (unsigned long long)before;
        const bool fits = s_base + (unsigned long long)total <= cap;  // workgroup-uniform
        if (lane == 0) oc[cell] = nolist ? make_int2(0, -1) : (fits ? make_int2((int)b64, n) : make_int2(0, 0));
        if (!fits || n == 0 || nolist) return;  // (no barrier below this line)
        base = (int)b64;
    }
    if (n <= kCellCap) {
        // rank sort: entry i goes to the number of entries that sort before it (ties by position)
        for (int i = lane; i < n; i += 64) {
            const int2 e = buf[i];
            const float key = __int_as_float(e.y);
            int rank = 0;
            for (int j = 0; j < n; j++) {
                const float kj = __int_as_float(buf[j].y);
                // (ties: by position; the camera's lists by PRIMITIVE INDEX -- each of their two parts, affine spheres and the rest, is
                // then in file order, which the ordered scan of a flagged primary ray merges: k_primary_cells)
                rank += (kj < key || (kj == key && (cam.on ? buf[j].x < e.x : j < i))) ? 1 : 0;
            }
            ent[(size_t)base + rank] = e;
        }
        return;
    }
    for (int i = lane; i < kCellCap; i += 64) ent[(size_t)base + i] = make_int2(buf[i].x, __float_as_int(-1.0f));
    }  // pass
}

// ------------------------------------------------------------------------------------------------
// Affine spheres (raytracing.cpp:571-576: only the DIRECTION of a shadow ray is mapped into object space, the origin
// stays where it is): which of them a shadow ray can pass depends on where along its line the ray starts, not only on
// the line.  The table holds, per (light, cell), the affine spheres that pass the strip test of k_affine_cells for the
// cell's rays (affine_strip_pass with the cell as the packet) and, for each, a 32-bit mask of the SLABS of origin
// distance D in which they still do: slab s = [s, s + 1] dmax / 32, widened by 1e-4 dmax either side; the strip
// packet of a slab has D in that range and origins within D_hi chord of the axis.  A ray looks at the entries of its
// cell whose mask has the bit of its own slab, runs filt_affine on those and hands the survivors to a pooled exact stage.
// ------------------------------------------------------------------------------------------------
constexpr int kAffSlabs = 32;

__device__ __forceinline__ StripPacket cell_strip(const CellCone& K, F3 light, float Da, float Db)
{
    StripPacket P;
    P.w = K.w;
    P.Lr = light;
    P.chord = K.sT / K.cLo * 1.0001f + 1e-6f;  // |ld - w| = 2 sin(theta / 2) <= tan(theta)
    P.Dmin = Da * K.cLo - 1e-3f;               // (L - o).w >= |L - o| cos(theta), less the origin bias
    P.Dmax = Db * 1.001f + 1e-3f;
    P.Dfar = Db * 1.001f + 1e-3f;
    P.rperp = Db * P.chord * 1.01f + 3e-4f;
    return P;
}

// One wave per (light, cell), one pass; the lists are per (light, cell, SLAB): oc is indexed by cell * kAffSlabs + slab and an entry
// is the index of the sphere's record in aff_shadow.  The slab masks of the first kAffKeep * 64 spheres are kept in LDS between the
// counting sweep and the storing sweep (the strip tests are what this kernel costs); spheres beyond that are tested again.
constexpr int kAffKeep = 16;
__global__ __launch_bounds__(64) void k_aff_build(SceneDev S, int G, const float* dmax, int2* oc, int* ent, unsigned long long* cursor,
                                                  unsigned long long cap)
{
    __shared__ unsigned keepm[kAffKeep][64];
    const int cell = blockIdx.x, lane = threadIdx.x;
    const int li = cell / (G * G), m = cell % (G * G);
    const CellCone K = cell_cone((unsigned)m, G);
    const F3 Lp = f3(S.lights[li].point);
    const float dm = dmax[li];
    const StripPacket K0 = cell_strip(K, Lp, 0.0f, dm);
    const float slab = dm * (1.0f / (float)kAffSlabs), eps = 1e-4f * dm;
    auto slabs_of = [&](const int ai, const bool valid) -> unsigned {
        const float4 m0 = S.aff_shadow[(size_t)ai * 4], m1 = S.aff_shadow[(size_t)ai * 4 + 1];
        const float4 m2 = S.aff_shadow[(size_t)ai * 4 + 2], m3 = S.aff_shadow[(size_t)ai * 4 + 3];
        unsigned mask = 0u;
        if (valid && affine_strip_pass(K0, m0, m1, m2, m3)) {
            // four slabs at a time first (a wider range of D contains its slabs' packets), then the slabs of the groups that pass
            for (int g = 0; g < kAffSlabs; g += 4) {
                const StripPacket Kg = cell_strip(K, Lp, fmaxf(0.0f, (float)g * slab - eps), (float)(g + 4) * slab + eps);
                if (!affine_strip_pass(Kg, m0, m1, m2, m3)) continue;
                for (int sl = g; sl < g + 4; sl++) {
                    const StripPacket Ks = cell_strip(K, Lp, fmaxf(0.0f, (float)sl * slab - eps), (float)(sl + 1) * slab + eps);
                    if (affine_strip_pass(Ks, m0, m1, m2, m3)) mask |= 1u << sl;
                }
            }
        }
        return mask;
    };
    int mine = 0;   // lane s: entries of slab s (sweep 0: counted; sweep 1: stored so far)
    int base = 0;   // lane s: where slab s's list starts
    for (int sweep = 0; sweep < 2; sweep++) {
        for (int a0 = 0, r = 0; a0 < S.naff; a0 += 64, r++) {
            const int ai = min(a0 + lane, S.naff - 1);
            unsigned mask;
            if (sweep == 1 && r < kAffKeep) mask = keepm[r][lane];
            else mask = slabs_of(ai, a0 + lane < S.naff);
            if (sweep == 0 && r < kAffKeep) keepm[r][lane] = mask;
            if (__ballot(mask != 0u) == 0ull) continue;  // wave-uniform
            for (int sl = 0; sl < kAffSlabs; sl++) {
                const bool in = (mask >> sl) & 1u;
                const unsigned long long bal = __ballot(in);
                if (bal == 0ull) continue;
                if (sweep == 1) {
                    const int before = __builtin_amdgcn_readlane(mine, sl), b0 = __builtin_amdgcn_readlane(base, sl);
                    if (in) ent[(size_t)b0 + before + lanes_below(bal)] = ai;
                }
                if (lane == sl) mine += __popcll(bal);
            }
        }
        if (sweep == 1) return;
        // the cell's 32 lists take one range: lane s's list starts at the range's start + the entries of the slabs before s
        int incl = (lane < kAffSlabs) ? mine : 0;
        for (int d = 1; d < kAffSlabs; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        const int total = __builtin_amdgcn_readlane(incl, kAffSlabs - 1);
        unsigned long long b64 = 0ull;
        if (lane == 0) b64 = atomicAdd(cursor, (unsigned long long)total);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));
        b64 = ((unsigned long long)hi << 32) | lo;
        const bool fits = b64 + (unsigned long long)total <= cap;  // wave-uniform
        base = (int)b64 + incl - mine;
        if (lane < kAffSlabs) oc[(size_t)cell * kAffSlabs + lane] = fits ? make_int2(base, mine) : make_int2(0, 0);
        if (!fits || total == 0) return;
        mine = 0;
        lds_fence();
    }
}

struct CellsLds {
    unsigned listS[kSpList];             // ray << 26 | sphere
    unsigned listT[kSpList];             // ray << 26 | primitive index of a triangle
    unsigned listA[kSpList];             // the same, for the area stage
    float listAt[kSpList];               // ... and the t of its plane intersection
    unsigned listX[kSpList];             // ray << 26 | affine sphere (primitive index)
    int hit[64];                         // 1: some primitive blocks ray r
};

// (light, cell) tables: plain shadow casters { primitive, dneed } and affine spheres { record, slab mask }.
// aoff == nullptr: no affine table (the caller runs k_shadow_affine behind the walk when the scene has affine spheres).
struct CellTab {
    const int2* off; const int2* ent; const float* dmax; int G;  // off[cell] = { first entry, entries } (the lists lie in build order)
    const int2* aoff; const int* aent; int aG;  // affine lists per (light, cell, slab), the same
    int nofilter;                                // A/B: entries go to the exact stages without the per-ray filter
};

// One lane = one shadow ray (o, ld) towards light li.  Returns true for the lanes whose ray is blocked by a plain
// primitive or -- with an affine table -- by an affine sphere.
template <bool EX>
__device__ __forceinline__ bool shadow_table_trace(const SceneDev& S, const CellTab& T, CellsLds& W, const bool live, const F3 o, const F3 ld,
                                                   const int li, Exec* ex)
{
    constexpr unsigned kPrimMask = (1u << kPoolPrimBits) - 1u;
    const int lane = threadIdx.x & 63;
    const F3 nld = normalized(ld);
    const bool abn = ray_abnormal(o, ld, nld, S.guard);
    const F3 Lp = f3(S.lights[li].point);
    const float ax = o.x - Lp.x, ay = o.y - Lp.y, az = o.z - Lp.z;
    const float D = fsqrt(ax * ax + ay * ay + az * az);  // as make_packet measures it
    const bool zero_dir = ld.x == 0.0f && ld.y == 0.0f && ld.z == 0.0f;
    const float dm = T.dmax[li];
    const bool covered = live && !abn && !zero_dir && D <= dm && D >= 1e-2f;
    if (live) ex_add(ex, EX_RAYS);
    int pos = 0, end = 0, posA = 0, endA = 0;
    if (covered) {
        const int cell = light_cell(nld, T.G);
        const int key = li * T.G * T.G + cell;
        const int2 oc = T.off[key];
        pos = oc.x;
        end = oc.x + oc.y;
        if (T.aoff) {
            const int cellA = (T.aG == T.G) ? cell : light_cell(nld, T.aG);
            const int slab = min(kAffSlabs - 1, (int)(D * ((float)kAffSlabs / dm)));
            const size_t keyA = ((size_t)li * T.aG * T.aG + cellA) * kAffSlabs + slab;
            const int2 oa = T.aoff[keyA];
            posA = oa.x;
            endA = oa.x + oa.y;
        }
    }
    W.hit[lane] = 0;
    lds_fence();
    const F3 fd = f3(nld.x * kDirScale, nld.y * kDirScale, nld.z * kDirScale);
    int nS = 0, nT = 0, nA = 0, nX = 0;

    // S / T / A: the pooled exact stages of shadow_plain_pool (rtx_shadow.hip.h), same tests on the same records
    auto run_S = [&](const int want) {
        const int cnt = min(want, nS);  // (never more than the list holds: a drain cannot index below it)
        const int b0 = nS - cnt;
        const bool has = lane < cnt;
        const unsigned e = W.listS[b0 + (has ? lane : 0)];
        const int k = (int)(e & kPrimMask), rr = (int)(e >> kPoolPrimBits);
        const F3 ro = f3(__shfl(o.x, rr), __shfl(o.y, rr), __shfl(o.z, rr));
        const F3 rn = f3(__shfl(nld.x, rr), __shfl(nld.y, rr), __shfl(nld.z, rr));
        const float4 hh = S.sph_hot[k];
        if (has) ex_add(ex, EX_SPH);
        float t;
        const bool hit = sphere_test(ro, rn, hh.x, hh.y, hh.z, hh.w, t);
        if (has && hit) W.hit[rr] = 1;
        nS = b0;
        lds_fence();
    };
    // X: exact affine-sphere tests (raytracing.cpp:571-582): direction through the inverse matrix, origin as it is
    auto run_X = [&](const int want) {
        const int cnt = min(want, nX);  // (never more than the list holds: a drain cannot index below it)
        const int b0 = nX - cnt;
        const bool has = lane < cnt;
        const unsigned e = W.listX[b0 + (has ? lane : 0)];
        const int k = (int)(e & kPrimMask), rr = (int)(e >> kPoolPrimBits);
        const F3 ro = f3(__shfl(o.x, rr), __shfl(o.y, rr), __shfl(o.z, rr));
        const F3 rl = f3(__shfl(ld.x, rr), __shfl(ld.y, rr), __shfl(ld.z, rr));
        const float4 hh = S.sph_hot[k];
        const F3 td = normalized(inv_apply(rl, S.sph_im[k]));
        if (has) ex_add(ex, EX_SPHX);
        float t;
        const bool hit = sphere_test(ro, td, hh.x, hh.y, hh.z, hh.w, t);
        if (has && hit) W.hit[rr] = 1;
        nX = b0;
        lds_fence();
    };
    auto run_A = [&](const int want) {
        const int cnt = min(want, nA);  // (never more than the list holds: a drain cannot index below it)
        const int b0 = nA - cnt;
        const bool has = lane < cnt;
        const unsigned e = W.listA[b0 + (has ? lane : 0)];
        const float t = W.listAt[b0 + (has ? lane : 0)];
        const int k = (int)(e & kPrimMask), rr = (int)(e >> kPoolPrimBits);
        const F3 ro = f3(__shfl(o.x, rr), __shfl(o.y, rr), __shfl(o.z, rr));
        const F3 rl = f3(__shfl(ld.x, rr), __shfl(ld.y, rr), __shfl(ld.z, rr));
        const float4* p = S.tri_hot + (size_t)(k - S.ns) * 5;
        const float4 q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4];
        if (has) ex_add(ex, EX_TRI_FULL);
        const bool hit = triangle_stage2(ro, rl, t, q1, q2, q3, q4);
        if (has && hit) W.hit[rr] = 1;
        nA = b0;
        lds_fence();
    };
    auto run_T = [&](const int want) {
        const int cnt = min(want, nT);  // (never more than the list holds: a drain cannot index below it)
        const int b0 = nT - cnt;
        const bool has = lane < cnt;
        const unsigned e = W.listT[b0 + (has ? lane : 0)];
        const int k = (int)(e & kPrimMask), rr = (int)(e >> kPoolPrimBits);
        const F3 ro = f3(__shfl(o.x, rr), __shfl(o.y, rr), __shfl(o.z, rr));
        const F3 rl = f3(__shfl(ld.x, rr), __shfl(ld.y, rr), __shfl(ld.z, rr));
        nT = b0;
        if (S.mt) {  // wave-uniform: RTX_F_TRI_MOLLER_TRUMBORE, one stage
            const TriRec R = load_tri(S.tri_hot, k - S.ns);
            float t, vd;
            if (has) ex_add(ex, EX_TRI_FULL);
            const bool hit = triangle_test_mt(ro, rl, R, t, vd) && !(t >= __builtin_inff());
            if (has && hit) W.hit[rr] = 1;
            lds_fence();
            return;
        }
        const float4* p = S.tri_hot + (size_t)(k - S.ns) * 5;
        const float4 q0 = p[0], q1 = p[1];
        const float4 e0 = S.tri_edge[(size_t)(k - S.ns) * 2], e1 = S.tri_edge[(size_t)(k - S.ns) * 2 + 1];
        float t = 0.0f;
        int exitp = 0;
        int need = triangle_stage1(ro, rl, q0, q1, e0, e1, t, EX ? &exitp : nullptr);
        if (!has) need = 0;
        if (EX && has && !need) ex_tri(ex, exitp);
        const unsigned long long bal = __ballot(need != 0);
        if (bal) {  // wave-uniform
            if (need) {
                const int p1 = nA + lanes_below(bal);
                W.listA[p1] = e;
                W.listAt[p1] = t;
            }
            nA += __popcll(bal);
        }
        lds_fence();
        if (nA >= 64) run_A(64);
    };

    // the walk: four entries of this lane's own list a step
    while (__any(pos < end)) {
        const bool act = pos < end;
        const int p0 = act ? pos : 0;
        const int2 e0 = T.ent[p0], e1 = T.ent[p0 + 1], e2 = T.ent[p0 + 2], e3 = T.ent[p0 + 3];
        const bool t0 = act && D >= __int_as_float(e0.y);
        const bool t1 = t0 && pos + 1 < end && D >= __int_as_float(e1.y);
        const bool t2 = t1 && pos + 2 < end && D >= __int_as_float(e2.y);
        const bool t3 = t2 && pos + 3 < end && D >= __int_as_float(e3.y);
        bool pass[4] = { t0, t1, t2, t3 };
        if (!T.nofilter) {  // kernel-uniform
            const float4 b0 = S.bounds[t0 ? e0.x : 0], b1 = S.bounds[t1 ? e1.x : 0], b2 = S.bounds[t2 ? e2.x : 0], b3 = S.bounds[t3 ? e3.x : 0];
            if (EX) ex_add(ex, EX_FILT, (unsigned)t0 + (unsigned)t1 + (unsigned)t2 + (unsigned)t3);
            pass[0] = t0 && filt_front(o, fd, b0); pass[1] = t1 && filt_front(o, fd, b1);
            pass[2] = t2 && filt_front(o, fd, b2); pass[3] = t3 && filt_front(o, fd, b3);
        }
        const int kk[4] = { e0.x, e1.x, e2.x, e3.x };
        pos = t3 ? pos + 4 : end;  // sorted by dneed: the first entry beyond this ray's distance ends its list
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool ps = pass[u] && kk[u] < S.ns, pt = pass[u] && kk[u] >= S.ns;
            const unsigned long long balS = __ballot(ps), balT = __ballot(pt);
            if ((balS | balT) == 0ull) continue;  // wave-uniform
            const unsigned entv = ((unsigned)lane << kPoolPrimBits) | (unsigned)kk[u];
            if (ps) W.listS[nS + lanes_below(balS)] = entv;
            if (pt) W.listT[nT + lanes_below(balT)] = entv;
            nS += __popcll(balS);
            nT += __popcll(balT);
            lds_fence();
            if (nS >= 64) run_S(64);
            if (nT >= 64) run_T(64);
        }
        if (W.hit[lane]) pos = end;
    }
    if (nT > 0) run_T(nT);
    if (nS > 0) run_S(nS);
    if (nA > 0) run_A(nA);

    // affine spheres, for the rays still lit: the list of the ray's own (light, cell, slab), mostly empty
    if (T.aoff) {  // kernel-uniform
        if (W.hit[lane]) posA = endA;
        while (__any(posA < endA)) {
            const bool act = posA < endA;
            const float4* r0 = S.aff_shadow + (size_t)T.aent[act ? posA : 0] * 4;
            const float4 a0 = r0[0], a1 = r0[1], a2 = r0[2], a3 = r0[3];
            if (EX && act) ex_add(ex, EX_AFILT);
            const bool pass = act && !(filt_affine(o, ld, a0, a1, a2, a3) < 0.0f);
            posA += 1;
            const unsigned long long bal = __ballot(pass);
            if (bal != 0ull) {  // wave-uniform
                if (pass) W.listX[nX + lanes_below(bal)] = ((unsigned)lane << kPoolPrimBits) | (unsigned)__float_as_int(a3.y);
                nX += __popcll(bal);
                lds_fence();
                if (nX >= 64) run_X(64);
                if (W.hit[lane]) posA = endA;
            }
        }
        if (nX > 0) run_X(nX);
    }
    bool done = W.hit[lane] != 0;

    // Rays the tables do not cover are traced one at a time, TRANSPOSED: the ray is broadcast and the 64 lanes take 64
    // cluster spheres of the shadow hierarchy a round (the per-ray cone test of k_closest_lanes, front half-space only:
    // the set holds plain primitives), then the 64 members of every passing cluster: filter and exact test; then -- with
    // an affine table -- 64 affine spheres a round.  An abnormal ray bypasses the bounds and takes every exact test.
    unsigned long long fallback = __ballot(live && !covered);
    while (fallback) {
        const int r = __ffsll((long long)fallback) - 1;
        fallback &= fallback - 1;
        auto bc = [&](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), r)); };
        const F3 ro = f3(bc(o.x), bc(o.y), bc(o.z)), rl = f3(bc(ld.x), bc(ld.y), bc(ld.z)), rn = f3(bc(nld.x), bc(nld.y), bc(nld.z));
        const F3 rfd = f3(rn.x * kDirScale, rn.y * kDirScale, rn.z * kDirScale);
        const bool rabn = __builtin_amdgcn_readlane((int)abn, r) != 0;
        bool blocked = false;
        for (int cb = 0; cb < S.sncl && !blocked; cb += 64) {
            const int c = cb + lane;
            bool cpass = false;
            if (c < S.sncl) {
                const float4 cs = S.scl[c];  // { c, -R^2 }
                if (!rabn) ex_add(ex, EX_CONE);
                cpass = rabn || !(cone_slack_front(ro, rn, make_float4(cs.x, cs.y, cs.z, fsqrt(-cs.w) * 1.00001f)) < 0.0f);
            }
            unsigned long long cm = __ballot(cpass);
            while (cm && !blocked) {
                const size_t mi = (size_t)(cb + (__ffsll((long long)cm) - 1)) * 64 + lane;
                cm &= cm - 1;
                const int k = S.sidx[mi];
                bool hit = false;
                if (k >= 0) {
                    if (!rabn) ex_add(ex, EX_FILT);
                    if (rabn || filt_front(ro, rfd, S.smem[mi])) hit = exact_any(S, k, ro, rl, rn, ex);
                }
                blocked = __any(hit);
            }
        }
        for (int a0 = 0; T.aoff && a0 < S.naff && !blocked; a0 += 64) {
            const int ai = a0 + lane;
            bool hit = false;
            if (ai < S.naff) {
                const float4 m0 = S.aff_shadow[(size_t)ai * 4], m1 = S.aff_shadow[(size_t)ai * 4 + 1];
                const float4 m2 = S.aff_shadow[(size_t)ai * 4 + 2], m3 = S.aff_shadow[(size_t)ai * 4 + 3];
                if (!rabn) ex_add(ex, EX_AFILT);
                if (rabn || !(filt_affine(ro, rl, m0, m1, m2, m3) < 0.0f)) hit = exact_any(S, __float_as_int(m3.y), ro, rl, rn, ex);
            }
            blocked = __any(hit);
        }
        if (lane == r && blocked) done = true;
    }
    return done;
}

// The shadow pass by table over a SORTED queue (k_shadow_bin / k_scan_* / k_shadow_scatter in front of it).
template <bool EX>
__global__ __launch_bounds__(kBlock) void k_shadow_cells(SceneDev S, CellTab T, ShadowQ Qi, const int* cnt_in, unsigned char* lit, int lit_cap)
{
    __shared__ CellsLds lds[kBlock / 64];
    const int n = *cnt_in;
    const int base = blockIdx.x * kBlock;
    if (base >= n) return;  // block-uniform
    Exec exs;
    Exec* ex = EX ? &exs : nullptr;
    ex_zero(ex);
    const int i = base + threadIdx.x;
    const int ii = (i < n) ? i : 0;
    const float4 qa = Qi.a[ii], qb = Qi.b[ii];
    const int id = __float_as_int(qb.z);
    const bool live = (i < n) && id >= 0;  // id < 0: alignment padding of the bin groups
    const bool done = shadow_table_trace<EX>(S, T, lds[threadIdx.x >> 6], live, f3(qa.x, qa.y, qa.z), f3(qa.w, qb.x, qb.y), live ? id / lit_cap : 0, ex);
    if (live && done) {
        lit[id] = 0;
        // no affine table: the ray leaves the queue, the affine kernel behind this one neither reads its lit byte nor traces it
        if (S.naff > 0 && !T.aoff) reinterpret_cast<int*>(Qi.b + i)[2] = -2;
    }
    ex_flush(ex, S.exec + 2 * EX_KINDS);
}

// The same straight off the hit records, no queue and no sort: thread = hit, blockIdx.y = light; writes every lit flag.
template <bool EX>
__global__ __launch_bounds__(kBlock) void k_shadow_direct(SceneDev S, CellTab T, LevelDev L, const int* nnodes)
{
    __shared__ CellsLds lds[kBlock / 64];
    const int n = min(*nnodes, L.cap);
    const int base = blockIdx.x * kBlock;
    if (base >= n) return;  // block-uniform
    Exec exs;
    Exec* ex = EX ? &exs : nullptr;
    ex_zero(ex);
    const int j = base + threadIdx.x, li = blockIdx.y;
    const bool live = j < n;
    const int jj = live ? j : 0;
    const rtx_light& Lt = S.lights[li];
    const F3 ip = f3(L.n_px[jj], L.n_py[jj], L.n_pz[jj]);
    const F3 nn = f3(L.n_nx[jj], L.n_ny[jj], L.n_nz[jj]);
    const F3 ld = normalized(f3(Lt.point.x - ip.x, Lt.point.y - ip.y, Lt.point.z - ip.z));        // :561-562
    const F3 o = f3(ip.x + nn.x * kBias, ip.y + nn.y * kBias, ip.z + nn.z * kBias);                // :568
    const bool done = shadow_table_trace<EX>(S, T, lds[threadIdx.x >> 6], live, o, ld, li, ex);
    if (live) L.lit[(size_t)li * L.cap + j] = done ? 0 : 1;
    ex_flush(ex, S.exec + 2 * EX_KINDS);
}

// ------------------------------------------------------------------------------------------------
// PRIMARY rays through the camera's table (round 3; k_cells_build in camera mode): one lane = one ray, which walks its cell's list
// and hands every entry to the pooled exact stage of k_closest_lanes (closest_pool's D and E: 64 (ray, primitive) pairs at a time,
// hits delivered to the lane that owns the ray and folded into its LexAcc, which does not depend on the order of arrival except that
// a ray's affine spheres come first -- they lead its list, and the stage is drained between the two walks).  No packet, no cluster
// hierarchy, no survivor broadcast: k_closest_f spent 50 ps a primary ray on C4 and 220 ps on C5 where the shadow rays, on their
// tables, take 26 and 48.  Rays the table does not cover (abnormal, a different origin, a cell without a list) take the ordered
// scan over every primitive, as the flagged rays of closest_pool do.
// ------------------------------------------------------------------------------------------------
struct CamTab { const int2* off; const int2* ent; int G; F3 apex; };

struct CamLds {
    float4 ray[3 * 64];        // origin (w: flags), normalised direction, direction as stored (w: the ray's nearest hit so far)
    unsigned listD[64 + 64];   // ray << 26 | primitive
    signed char slot[64];
    int inK[64];
    float inT[64];
};

template <bool EX>
__global__ __launch_bounds__(kBlock) void k_primary_cells(SceneDev S, CamTab T, LevelDev L, const int* nrays)
{
    __shared__ CamLds lds[kBlock / 64];
    const int n = min(*nrays, L.cap);
    const int base = blockIdx.x * kBlock;
    if (base >= n) return;  // block-uniform
    Exec exs;
    Exec* ex = EX ? &exs : nullptr;
    ex_zero(ex);
    CamLds& W = lds[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const int q = base + threadIdx.x;
    const bool inq = q < n;
    const int i = inq ? q : 0;
    const F3 o = f3(L.ox[i], L.oy[i], L.oz[i]);
    const F3 d = f3(L.dx[i], L.dy[i], L.dz[i]);
    const F3 nd = normalized(d);
    const bool live = inq && !ray_dead(o, d);
    const bool abn = live && ray_abnormal(o, d, nd, S.guard);
    const bool zero_dir = d.x == 0.0f && d.y == 0.0f && d.z == 0.0f;
    if (lane == 0) ex_add(ex, EX_WAVES);
    if (inq) ex_add(ex, EX_RAYS);
    float4* const rayO = W.ray; float4* const rayN = W.ray + 64; float4* const rayD = W.ray + 128;
    rayO[lane] = make_float4(o.x, o.y, o.z, 0.0f);
    rayN[lane] = make_float4(nd.x, nd.y, nd.z, 0.0f);
    rayD[lane] = make_float4(d.x, d.y, d.z, kMaxT);
    W.slot[lane] = -1;
    // the ray's list
    int pos = 0, end = 0;
    bool covered = live && !abn && !zero_dir && o.x == T.apex.x && o.y == T.apex.y && o.z == T.apex.z;
    if (covered) {
        const int2 oc = T.off[light_cell(nd, T.G)];
        pos = oc.x; end = oc.x + oc.y;
        if (oc.y < 0) { covered = false; end = pos; }
    }
    const int beg = pos;
    int mid = pos;  // where the list's affine spheres end (known after walk 0)
    lds_fence();
    LexAcc h;
    h.nearest = kMaxT; h.best = -1; h.overflow = false;
#pragma unroll
    for (int e = 0; e < kAffList; e++) { h.ak[e] = -1; h.at[e] = 0.0f; }
    int nD = 0;
    auto deliver = [&](bool pending, int rr, int k, float t, int kind) {
        while (__any(pending)) {
            if (pending) W.slot[rr] = (signed char)lane;
            lds_fence();
            const bool win = pending && W.slot[rr] == lane;
            if (win) { W.inK[rr] = k | (kind << 28); W.inT[rr] = t; pending = false; }
            lds_fence();
            if (W.slot[lane] >= 0) {
                const int e = W.inK[lane];
                lex_update(h, e & ((1 << 28) - 1), W.inT[lane], e >> 28);
                W.slot[lane] = -1;
                rayD[lane].w = h.nearest;
            }
            lds_fence();
        }
    };
    auto run_D = [&](const int want) {
        const int cnt = min(want, nD), b0 = nD - cnt;
        const bool has = lane < cnt;
        const unsigned e = W.listD[b0 + (has ? lane : 0)];
        const int k = (int)(e & ((1u << kPoolPrimBits) - 1u));
        const int rr = (int)(e >> kPoolPrimBits);
        const float4 rO = rayO[rr], rN = rayN[rr], rD = rayD[rr];
        float t = 0.0f;
        int kind = 0;
        // (rD.w: the ray's nearest hit as its owner last wrote it: a triangle matters only at t < nearest, or at t == nearest with a
        // lower index -- everything at t >= the float above nearest leaves early)
        if (has) kind = exact_hit_edge(S, k, f3(rO.x, rO.y, rO.z), f3(rD.x, rD.y, rD.z), f3(rN.x, rN.y, rN.z), t, __int_as_float(__float_as_int(rD.w) + 1), ex);
        nD = b0;
        deliver(kind != 0, rr, k, t, kind);
    };
    const int kAffKey = __float_as_int(-2.0f);
    for (int walk = 0; walk < 2; walk++) {
        // walk 0: the affine spheres at the head of every list; walk 1: the rest
        for (;;) {
            int2 e = make_int2(-1, 0);
            if (pos < end) e = T.ent[pos];
            const bool take = pos < end && e.x >= 0 && (walk == 1 || e.y == kAffKey);
            const unsigned long long bal = __ballot(take);
            if (bal == 0ull) break;  // wave-uniform
            if (take) { W.listD[nD + lanes_below(bal)] = ((unsigned)lane << kPoolPrimBits) | (unsigned)e.x; pos++; }
            nD += __popcll(bal);
            lds_fence();
            if (nD >= 64) run_D(64);
        }
        while (nD > 0) run_D(nD);
        if (walk == 0) mid = pos;
    }
    bool aliveList = false;
#pragma unroll
    for (int e = 0; e < kAffList; e++) aliveList = aliveList || h.ak[e] >= 0;
    float nearest = h.nearest;
    int best = h.best;
    int tflag = (best >= 0 && best < S.ns && aliveList) ? kTFlag : 0;
    // A flagged ray WITH a list (its affine-sphere hits overflowed LexAcc) takes the reference's ordered scan over that list: it holds
    // every primitive whose exact test can accept the ray, its two parts each in file order (k_cells_build), merged here by index.
    // Neighbouring primary rays overflow together -- whole waves in a scene whose affine spheres overlap five deep -- and the scan
    // over every primitive cost such a wave 10,000 steps where its lists hold some forty entries (perf_survey.py "dense": the
    // primary stage 1.99 -> 0.33 ms, the frame 3.29 -> 1.58).
    const bool slow_list = live && covered && h.overflow;
    if (__any(slow_list)) {
        HitAcc hs;
        hs.nearest = kMaxT; hs.best = -1; hs.tflag = 0; hs.txf = kMaxT;
        int pa = beg, pb = mid;
        while (__any(slow_list && (pa < mid || pb < end))) {
            if (slow_list && (pa < mid || pb < end)) {
                const int ka = (pa < mid) ? T.ent[pa].x : 0x7fffffff, kb = (pb < end) ? T.ent[pb].x : 0x7fffffff;
                const int k = min(ka, kb);
                if (ka <= kb) pa++; else pb++;
                if (k >= 0 && k != 0x7fffffff) exact_closest(S, k, o, d, nd, hs, ex);
            }
        }
        if (slow_list) { nearest = hs.nearest; best = hs.best; tflag = hs.tflag; }
    }
    const bool slow = live && !covered;
    ordered_scan(S, slow, abn, o, d, nd, nearest, best, tflag, ex);
    if (S.pstat) {  // RTX_F_PACKET_STATS only
        const unsigned long long sl = __ballot(slow_list), su = __ballot(slow);
        if (lane == 0 && (sl | su)) atomicAdd(S.pstat + 2, (unsigned long long)__popcll(sl) + ((unsigned long long)__popcll(su) << 32));
    }
    if (inq) {
        L.t[i] = nearest;
        L.prim[i] = (best < 0) ? -1 : (best | tflag);
    }
    ex_flush(ex, S.exec + 0 * EX_KINDS);
}

}  // namespace rtx
#endif
