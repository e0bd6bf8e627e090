// rtx_shade.hip.h -- hit set-up and spawn, exact-only shadow kernel, unwind (shading), antialias + quantise
// Part of the one translation unit rtx_device.hip (device code, namespace rtx); not a stand-alone header.
#ifndef RTX_SHADE_HIP_H_
#define RTX_SHADE_HIP_H_

namespace rtx {

// ------------------------------------------------------------------------------------------------
// K3 hit set-up and spawn (raytracing.cpp:436-526)
// ------------------------------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_setup(SceneDev S, LevelDev L, LevelDev N, const int* nrays,
                                                 int* nnodes, int spawn, int* overflow)  // nnodes[1] = the rays of the next generation
{
    int i = blockIdx.x * BLOCK + threadIdx.x;
    bool live = i < min(*nrays, L.cap);
    int pw = live ? L.prim[i] : -1;
    bool hit = pw >= 0;
    if (live && !hit) {  // :436-438: background, no ambient
        L.val[3 * i + 0] = 0.4f; L.val[3 * i + 1] = 0.4f; L.val[3 * i + 2] = 0.4f;
    }
    // ONE allocation per workgroup for the node slot and both kinds of child (round 3; the node and the children were two
    // allocations, seven barriers between them, each waiting on the slowest of the workgroup's waves): whether a hit spawns a
    // reflection or a refraction ray is a property of the primitive's material alone, read here ahead of the geometry.
    constexpr int nw = BLOCK / 64;
    __shared__ int sh_alloc[3 * nw + 2];
    bool want_refl = false, want_refr = false;
    if (hit && spawn) {  // depth < MAX_depth (:449)
        const float4 m0 = S.mat4[(size_t)(pw & ~kTFlag) * 4];  // { reflectivity, transparency, refractive index, radius }
        want_refl = m0.x > 0.0f;
        want_refr = m0.y > 0.0f;
    }
    int node = -1, crefl = -1, crefr = -1;
    {
        const unsigned long long m0 = __ballot(hit), m1 = __ballot(want_refl), m2 = __ballot(want_refr);
        const int w = threadIdx.x >> 6;
        const int p0 = __builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0));
        const int p1 = __builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0));
        const int p2 = __builtin_amdgcn_mbcnt_hi((unsigned)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m2, 0));
        if ((threadIdx.x & 63) == 0) { sh_alloc[w] = __popcll(m0); sh_alloc[nw + w] = __popcll(m1); sh_alloc[2 * nw + w] = __popcll(m2); }
        __syncthreads();
        int off0 = 0, off1 = 0, off2 = 0, tot0 = 0, tot1 = 0, tot2 = 0;
        for (int k = 0; k < nw; k++) {
            if (k < w) { off0 += sh_alloc[k]; off1 += sh_alloc[nw + k]; off2 += sh_alloc[2 * nw + k]; }
            tot0 += sh_alloc[k]; tot1 += sh_alloc[nw + k]; tot2 += sh_alloc[2 * nw + k];
        }
        if (threadIdx.x == 0) {
            // ONE atomic for both counters: the nodes of this generation and the rays of the next are the two halves of an aligned
            // 64-bit word (rtx_device.hip: d_cnt).  Atomics on one address are served one at a time, ~12 ns each: generation 0 of
            // a 4K frame has 8,100 workgroups.  (Slots of sh_alloc the loop above does not read; no carry: counts stay below 2^31.)
            const unsigned long long add = (unsigned long long)(unsigned)tot0 | ((unsigned long long)(unsigned)(tot1 + tot2) << 32);
            const unsigned long long old = add ? atomicAdd(reinterpret_cast<unsigned long long*>(nnodes), add) : 0ull;
            sh_alloc[3 * nw] = (int)(unsigned)old;
            sh_alloc[3 * nw + 1] = (int)(unsigned)(old >> 32);
        }
        __syncthreads();
        const int base0 = sh_alloc[3 * nw], base1 = sh_alloc[3 * nw + 1];
        if (hit) node = base0 + off0 + p0;
        // reflection rays take the first slots of the workgroup's run, refraction rays the rest
        if (want_refl) crefl = base1 + off1 + p1;
        if (want_refr) crefr = base1 + tot1 + off2 + p2;
        // never past the next level's capacity (measured at creation, rtx_device.hip: cannot happen for the frame the
        // context was calibrated on): the child is dropped and counted, its colour term is then simply missing
        if (want_refl && crefl >= N.cap) { want_refl = false; crefl = -1; atomicAdd(overflow, 1); }
        if (want_refr && crefr >= N.cap) { want_refr = false; crefr = -1; atomicAdd(overflow, 1); }
    }
    F3 o = f3(0, 0, 0), d = o, normal = o, ip = o, rd = o;
    float fres = 0.0f, fratio = 0.0f, eta = 1.0f;
    int srcp = -1;  // a PLAIN sphere: its children are tested against it first (LevelDev::src)
    if (hit) {
        o = f3(L.ox[i], L.oy[i], L.oz[i]);
        d = f3(L.dx[i], L.dy[i], L.dz[i]);
        float nearest = L.t[i];
        bool tflag = (pw & kTFlag) != 0;
        int prim = pw & ~kTFlag;
        bool is_sphere = prim < S.ns;
        const float4 m0 = S.mat4[(size_t)prim * 4];  // { reflectivity, transparency, refractive index, radius }: the records' own floats
        const float reflectivity = m0.x, transparency = m0.y, ri = m0.z;
        if (is_sphere) {
            const float4 sh = S.sph_hot[prim];  // { centre, r * r }
            const bool sxf = S.sph_xf[prim] != 0;
            if (!sxf) srcp = prim;
            // recompute the winner's object-space normal (:229-231, :395-398)
            F3 to = o, td = d;
            if (sxf) { const rtx_xmatrix& im = S.sph_im[prim]; to = inv_apply(o, im); td = inv_apply(d, im); }
            td = normalized(td);
            normal = f3((to.x + td.x * nearest - sh.x) / m0.w, (to.y + td.y * nearest - sh.y) / m0.w, (to.z + td.z * nearest - sh.z) / m0.w);  // sphere_normal
            if (sxf) normal = inv_apply_normal(normal, S.sph_im[prim]);
        } else if (prim < S.ns + S.nt) {
            const float4* p = S.tri_hot + (size_t)(prim - S.ns) * 5;
            float4 q0 = p[0];
            float vd = q0.x * d.x + q0.y * d.y + q0.z * d.z;
            normal = (vd > 0.0f) ? f3(-q0.x, -q0.y, -q0.z) : f3(q0.x, q0.y, q0.z);  // :343-347
        } else {
            const float4 q0 = S.phot[(size_t)prim * 5];
            normal = f3(q0.x, q0.y, q0.z);  // :311: the stored normal, negated
            if (__float_as_int(S.phot[(size_t)prim * 5 + 1].z) == -1) {
                // a plane (RTX_F_TRACE_PLANES; intersectplane returns no normal): the triangle's facing rule, :343-347
                const float vd = q0.x * d.x + q0.y * d.y + q0.z * d.z;
                if (vd > 0.0f) normal = f3(-q0.x, -q0.y, -q0.z);
            }
        }
        ip = f3(o.x + nearest * d.x, o.y + nearest * d.y, o.z + nearest * d.z);  // :440
        if (tflag) ip = fwd_apply(ip, S.sph[prim].mat);                          // :441-444
        normal = normalized(normal);                                             // :445
        if (spawn) {  // depth < MAX_depth (:449)
            bool insphere = false;
            rd = d;
            if (is_sphere) {
                if (tflag) rd = inv_apply(d, S.sph_im[prim]);  // :452-455
                rd = normalized(rd);
                if (dot3(normal, rd) > 0.0f) {                 // :458-461
                    normal = f3(-normal.x, -normal.y, -normal.z);
                    insphere = true;
                }
            }
            // (want_refl / want_refr: reflectivity > 0 / transparency > 0, decided before the allocation above; a child that found
            // no room is dropped there)
            if (reflectivity > 0.0f || transparency > 0.0f) {
                fratio = -(dot3(rd, normal));
                fres = fresnel_term(fratio);
                eta = ri;
                if (insphere) eta = (float)(1.0 / (double)ri);  // :477-479 (spheres only)
            }
        }
        L.n_ray[node] = i;
        L.n_nx[node] = normal.x; L.n_ny[node] = normal.y; L.n_nz[node] = normal.z;
        L.n_px[node] = ip.x; L.n_py[node] = ip.y; L.n_pz[node] = ip.z;
        L.n_fres[node] = fres;
    }
    if (hit) { L.n_crefl[node] = crefl; L.n_crefr[node] = crefr; }
    if (want_refl) {  // :464-468
        F3 r = f3(rd.x + 2.0f * fratio * normal.x, rd.y + 2.0f * fratio * normal.y, rd.z + 2.0f * fratio * normal.z);
        r = normalized(r);
        N.ox[crefl] = ip.x + normal.x * kBias; N.oy[crefl] = ip.y + normal.y * kBias; N.oz[crefl] = ip.z + normal.z * kBias;
        N.dx[crefl] = r.x; N.dy[crefl] = r.y; N.dz[crefl] = r.z;
        N.src[crefl] = srcp;
    }
    if (want_refr) {  // :476-488
        float raydotn = fratio;
        float k = (float)(1.0 - (double)(eta * eta) * (1.0 - (double)(raydotn * raydotn)));
        float coef = eta * raydotn - sqrtf(k);  // NaN when k < 0: the child then misses everything
        F3 r = f3(rd.x * eta + normal.x * coef, rd.y * eta + normal.y * coef, rd.z * eta + normal.z * coef);
        r = normalized(r);
        N.ox[crefr] = ip.x - normal.x * kBias; N.oy[crefr] = ip.y - normal.y * kBias; N.oz[crefr] = ip.z - normal.z * kBias;
        N.dx[crefr] = r.x; N.dy[crefr] = r.y; N.dz[crefr] = r.z;
        N.src[crefr] = srcp;
    }
}

// ------------------------------------------------------------------------------------------------
// K4 shadow any-hit (raytracing.cpp:560-589): one thread per (hit, light); blockIdx.y = light.
// No distance limit: any primitive at t >= 0 along the ray blocks the light.
// ------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_shadow(SceneDev S, LevelDev L, const int* nnodes, unsigned long long* tests)
{
    int j = blockIdx.x * kBlock + threadIdx.x;
    bool live = j < *nnodes;
    if (!COUNT && !live) return;
    unsigned int cnt[4] = {0, 0, 0, 0};
    if (live) {
    int li = blockIdx.y;
    F3 ip = f3(L.n_px[j], L.n_py[j], L.n_pz[j]);
    F3 n = f3(L.n_nx[j], L.n_ny[j], L.n_nz[j]);
    const rtx_light& Lt = S.lights[li];
    F3 ld = normalized(f3(Lt.point.x - ip.x, Lt.point.y - ip.y, Lt.point.z - ip.z));  // :561-562
    F3 o = f3(ip.x + n.x * kBias, ip.y + n.y * kBias, ip.z + n.z * kBias);            // :568
    F3 nld = normalized(ld);  // :575 for untransformed spheres
    bool shadow = false;
    for (int s = 0; s < S.ns && !shadow; s++) {
        float4 h = S.sph_hot[s];
        F3 td = nld;
        if (S.sph_xf[s]) td = normalized(inv_apply(ld, S.sph_im[s]));  // :572-575; origin NOT transformed
        float t;
        if (COUNT) cnt[S.sph_xf[s] ? 1 : 0]++;
        shadow = sphere_test(o, td, h.x, h.y, h.z, h.w, t);
    }
    for (int k = 0; k < S.nt && !shadow; k++) {
        TriRec T = load_tri(S.tri_hot, k);
        float t = 0.0f, vd;
        shadow = triangle_test(o, ld, T, t, vd, nullptr, S.mt != 0);
        if (COUNT) cnt[(vd == 0.0f || t < 0.0f) ? 3 : 2]++;
    }
    L.lit[(size_t)li * L.cap + j] = shadow ? 0 : 1;
    }
    if (COUNT) flush_counts(cnt, tests);
}

// ------------------------------------------------------------------------------------------------
// K5 unwind (raytracing.cpp:446, :470-472, :490-492, :590-646): colour of one hit from its
// children's colours, in the reference's add order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_unwind(SceneDev S, LevelDev L, LevelDev N, const int* nnodes)
{
    int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= *nnodes) return;
    int i = L.n_ray[j];
    int pw = L.prim[i];
    int prim = pw & ~kTFlag;
    bool is_sphere = prim < S.ns;
    F3 d = f3(L.dx[i], L.dy[i], L.dz[i]);
    F3 n = f3(L.n_nx[j], L.n_ny[j], L.n_nz[j]);
    F3 ip = f3(L.n_px[j], L.n_py[j], L.n_pz[j]);
    // the material, packed (rtx_pack.h: mat4; polygons take the triangle forms with their own material, :527-556, :627-639)
    const float4 m0 = S.mat4[(size_t)prim * 4], m1 = S.mat4[(size_t)prim * 4 + 1], m2 = S.mat4[(size_t)prim * 4 + 2], m3 = S.mat4[(size_t)prim * 4 + 3];
    const F3 color = f3(m1.x, m1.y, m1.z), spec = f3(m2.x, m2.y, m2.z), diff = f3(m3.x, m3.y, m3.z);
    const float transparency = m0.y, specexpo = m1.w;
    const bool xf = is_sphere && m2.w != 0.0f;
    F3 pix = f3(0.0f, 0.0f, 0.0f);
    float fres = L.n_fres[j];
    int crefl = L.n_crefl[j], crefr = L.n_crefr[j];
    if (crefl >= 0) {  // :470-472
        pix.x += N.val[3 * crefl + 0] * fres * color.x;
        pix.y += N.val[3 * crefl + 1] * fres * color.y;
        pix.z += N.val[3 * crefl + 2] * fres * color.z;
    }
    if (crefr >= 0) {  // :490-492: (1.0 - fresnel) makes the whole right-hand side double
        double w = 1.0 - (double)fres;
        pix.x = (float)((double)pix.x + (double)N.val[3 * crefr + 0] * w * (double)transparency * (double)color.x);
        pix.y = (float)((double)pix.y + (double)N.val[3 * crefr + 1] * w * (double)transparency * (double)color.y);
        pix.z = (float)((double)pix.z + (double)N.val[3 * crefr + 2] * w * (double)transparency * (double)color.z);
    }
    // view direction used by the specular term (:592-599 spheres, :619 triangles)
    F3 vdir = d;
    if (is_sphere) {
        if (xf) vdir = inv_apply(d, S.sph_im[prim]);
        vdir = normalized(vdir);
    }
    for (int li = 0; li < S.nl; li++) {
        if (!L.lit[(size_t)li * L.cap + j]) continue;
        const rtx_light& Lt = S.lights[li];
        F3 lc = f3(Lt.color);
        F3 ld = normalized(f3(Lt.point.x - ip.x, Lt.point.y - ip.y, Lt.point.z - ip.z));
        float dp = dot3(n, ld);
        float dotprod = (0.0f < dp) ? dp : 0.0f;  // std::max(float(0.0), dp): NaN -> 0
        F3 ldir = ld;
        if (is_sphere) {
            if (xf) ldir = inv_apply(ld, S.sph_im[prim]);
            ldir = normalized(ldir);
        }
        float ndoti2 = 2.0f * (dot3(ldir, n));
        F3 rr = normalized(f3(ndoti2 * n.x - ldir.x, ndoti2 * n.y - ldir.y, ndoti2 * n.z - ldir.z));
        float rdotv = -(dot3(vdir, rr));
        float pw_ = powf_rn(rdotv, specexpo);
        pix.x += spec.x * pw_ * lc.x;
        pix.y += spec.y * pw_ * lc.y;
        pix.z += spec.z * pw_ * lc.z;
        pix.x += dotprod * diff.x * lc.x;
        pix.y += dotprod * diff.y * lc.y;
        pix.z += dotprod * diff.z * lc.z;
    }
    pix.x += S.ambient.x; pix.y += S.ambient.y; pix.z += S.ambient.z;  // :643-645
    L.val[3 * i + 0] = pix.x; L.val[3 * i + 1] = pix.y; L.val[3 * i + 2] = pix.z;
}

// ------------------------------------------------------------------------------------------------
// K6 3x3 antialias + quantise (raytracing.cpp:745-780, :792-794).  The reference re-traces the
// neighbour rays; raytrace() is a pure function of the pixel, so filtering the 1-spp image in
// the same row-major add order is byte-identical.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_aa_quantise(const float* fimg, const int* row2ti, const int* own_rows,
                                                        int nown, int width, int height, unsigned char* out)
{
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nown * width) return;
    int r = i / width, x = i - r * width;
    int y = own_rows[r];
    float counter = 0.0f, ax = 0.0f, ay = 0.0f, az = 0.0f;
    for (int yv = y - 1; yv <= y + 1; yv++)
        for (int xv = x - 1; xv <= x + 1; xv++)
            if (xv >= 0 && yv >= 0 && xv < width && yv < height) {
                const float* p = fimg + ((size_t)row2ti[yv] * width + xv) * 3;
                counter = counter + 1.0f;
                if (yv == y && xv == x) { ax += 2.0f * p[0]; ay += 2.0f * p[1]; az += 2.0f * p[2]; }
                else { ax += p[0]; ay += p[1]; az += p[2]; }
            }
    float dv = counter + 1.0f;
    ax = ax / dv; ay = ay / dv; az = az / dv;
    out[3 * (size_t)i + 0] = quantise(ax);
    out[3 * (size_t)i + 1] = quantise(ay);
    out[3 * (size_t)i + 2] = quantise(az);
}

// The same for FOUR neighbouring pixels per thread (width a multiple of 4, output dword-aligned): the 3 x 6 window
// comes in with three 16-byte loads per row instead of 27 dword loads per pixel, the 12 output bytes leave as three
// dwords.  Every pixel's sum is formed in the reference's order (rows top to bottom, left to right, the centre
// doubled), so the bytes are those of k_aa_quantise.
__global__ __launch_bounds__(kBlock) void k_aa_quantise4(const float* fimg, const int* row2ti, const int* own_rows,
                                                         int nown, int width, int height, unsigned char* out)
{
    const int q = blockIdx.x * kBlock + threadIdx.x;       // group of 4 pixels
    const int wq = width >> 2;
    if (q >= nown * wq) return;
    const int r = q / wq, x0 = (q - r * wq) << 2;
    const int y = own_rows[r];
    float win[3][6][3];
    bool rowok[3];
#pragma unroll
    for (int dy = 0; dy < 3; dy++) {
        const int yv = y - 1 + dy;
        rowok[dy] = yv >= 0 && yv < height;
        const float* p = fimg + ((size_t)row2ti[rowok[dy] ? yv : y] * width + x0) * 3;
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4), c = *reinterpret_cast<const float4*>(p + 8);
        const float m[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) win[dy][k + 1][ch] = m[k * 3 + ch];
        // the pixel on either side: one aligned 16-byte load each (p is 48-byte aligned), not three dwords
        float4 lft = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rgt = lft;
        if (x0 > 0) lft = *reinterpret_cast<const float4*>(p - 4);
        if (x0 + 4 < width) rgt = *reinterpret_cast<const float4*>(p + 12);
        win[dy][0][0] = lft.y; win[dy][0][1] = lft.z; win[dy][0][2] = lft.w;
        win[dy][5][0] = rgt.x; win[dy][5][1] = rgt.y; win[dy][5][2] = rgt.z;
    }
    unsigned char bytes[12];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = x0 + k;
        float counter = 0.0f, ax = 0.0f, ay = 0.0f, az = 0.0f;
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                const int xv = x - 1 + dx;
                if (rowok[dy] && xv >= 0 && xv < width) {
                    const float* pv = win[dy][k + dx];
                    counter = counter + 1.0f;
                    if (dy == 1 && dx == 1) { ax += 2.0f * pv[0]; ay += 2.0f * pv[1]; az += 2.0f * pv[2]; }
                    else { ax += pv[0]; ay += pv[1]; az += pv[2]; }
                }
            }
        const float dv = counter + 1.0f;
        bytes[3 * k + 0] = quantise(ax / dv);
        bytes[3 * k + 1] = quantise(ay / dv);
        bytes[3 * k + 2] = quantise(az / dv);
    }
    unsigned int* o = reinterpret_cast<unsigned int*>(out + 12 * (size_t)q);
#pragma unroll
    for (int j = 0; j < 3; j++)
        o[j] = (unsigned)bytes[4 * j] | ((unsigned)bytes[4 * j + 1] << 8) | ((unsigned)bytes[4 * j + 2] << 16) | ((unsigned)bytes[4 * j + 3] << 24);
}

// counters of one chunk -> 64-bit frame totals
__global__ void k_accumulate(const int* cnt, unsigned long long* totals, int levels)
{
    int k = threadIdx.x;  // totals: rays per level, then nodes per level; cnt: rays of level k at [2k + 1], nodes at [2k + 2]
    if (k < 2 * levels) totals[k] += (unsigned long long)cnt[k < levels ? 2 * k + 1 : 2 * (k - levels) + 2];
}


}  // namespace rtx
#endif
