// dsdf_bsdf.h -- the `principled` BSDF of sdf_direct_reparam's principled-* configs (python/opt_configs.py:288-299: the
// optimised volumes are 'main-bsdf.base_color.volume.data' and 'main-bsdf.roughness.volume.data').  Host/device inline.
//
// The plugin is third-party (Mitsuba 3 src/bsdfs/principled.cpp, principled_helpers.h, microfacet.h, fresnel.h; absent from
// the reference repository, which ships no scene files either).  The parameters the configs do not optimise sit at the
// plugin's defaults -- metallic 0, specular 0.5 (eta 1.5), spec_tint 0, spec_trans 0, anisotropic 0, sheen 0, clearcoat 0,
// flatness 0 -- which leaves two lobes: diffuse + retro-reflection, and the main specular reflection (GGX, separable Smith
// G1, unpolarised dielectric Fresnel).  Without anisotropy every term is a function of four scalars
//     x = n . wi,   y = n . wo,   u = wi . wo,   r = roughness(p)
// (half vector h = (wi + wo) / |wi + wo|:  wi . h = wo . h = sqrt((1 + u) / 2),  n . h = (x + y) / (2 wi . h)), so
//     eval(si, wo) * |cos theta_o| = base_color * Kd(x, y, u, r) + Ks(x, y, u, r)
// and the adjoint needs the eight partial derivatives below.  The test suite restates the plugin in the local shading frame,
// vector by vector, and checks value and partials against the autograd of that restatement (tests/test_principled_host.py).
#pragma once

#define DSDF_PRINCIPLED_ETA 1.5f          /* principled.cpp: eta = 2 / (1 - sqrt(0.08 * specular)) - 1, specular = 0.5 */

struct PrincipledTerms { float kd, ks, dkd[4], dks[4]; };     // partials w.r.t. (x, y, u, r)

// Both cosines positive (the caller's `front` test: reflect && front_side of Principled::eval).
DSDF_HD PrincipledTerms principled_terms(float x, float y, float u, float r) {
    PrincipledTerms T;
    const float inv_pi = 0.3183098861837907f;
    // ---- diffuse + retro-reflection: |cos_o| / pi * base_color * (f_diff + f_retro)
    const float mx = 1.f - x, my = 1.f - y;
    const float mx2 = mx * mx, my2 = my * my;
    const float Fi = mx2 * mx2 * mx, Fo = my2 * my2 * my;                 // schlick_weight
    const float dFi = -5.f * mx2 * mx2, dFo = -5.f * my2 * my2;
    const float Rr = r * (1.f + u);                                      // 2 r cos_theta_d^2, cos_theta_d = wo . h
    const float fd = (1.f - 0.5f * Fi) * (1.f - 0.5f * Fo);
    const float mix = Fo + Fi + Fo * Fi * (Rr - 1.f);
    const float fr = Rr * mix;
    T.kd = y * inv_pi * (fd + fr);
    T.dkd[0] = y * inv_pi * (-0.5f * dFi * (1.f - 0.5f * Fo) + Rr * (dFi + Fo * dFi * (Rr - 1.f)));
    T.dkd[1] = inv_pi * (fd + fr) + y * inv_pi * (-0.5f * dFo * (1.f - 0.5f * Fi) + Rr * (dFo + dFo * Fi * (Rr - 1.f)));
    const float dkd_dRr = y * inv_pi * (mix + Rr * Fo * Fi);
    T.dkd[2] = dkd_dRr * r;
    T.dkd[3] = dkd_dRr * (1.f + u);
    // ---- main specular reflection: F D G / (4 |cos_i|)
    const float alpha = fmaxf(r * r, 0.001f);                            // calc_dist_params, no anisotropy
    const float a2 = alpha * alpha;
    const float da2_dr = r * r > 0.001f ? 4.f * r * r * r : 0.f;
    const float c = sqrtf(fmaxf(0.5f * (1.f + u), 1e-12f));              // wi . h
    const float ch = (x + y) / (2.f * c);                                // n . h
    // GGX: D = a2 / (pi s^2), s = 1 - ch^2 + a2 ch^2
    const float s = 1.f - ch * ch + a2 * ch * ch;
    const float D = a2 * inv_pi / (s * s);
    const float dlnD_dch = -4.f * ch * (a2 - 1.f) / s;
    const float dlnD_da2 = 1.f / a2 - 2.f * ch * ch / s;
    // Smith G1(v) = 2 / (1 + q), q = sqrt(1 + a2 (1 / v^2 - 1))
    const float qx = sqrtf(1.f + a2 * (1.f / (x * x) - 1.f)), qy = sqrtf(1.f + a2 * (1.f / (y * y) - 1.f));
    const float Gx = 2.f / (1.f + qx), Gy = 2.f / (1.f + qy);
    const float dlnGx_dx = a2 / ((1.f + qx) * qx * x * x * x), dlnGy_dy = a2 / ((1.f + qy) * qy * y * y * y);
    const float dlnGx_da2 = -(1.f / (x * x) - 1.f) / (2.f * (1.f + qx) * qx), dlnGy_da2 = -(1.f / (y * y) - 1.f) / (2.f * (1.f + qy) * qy);
    // dielectric Fresnel at cos = c >= 0, outside: eta_it = eta
    const float eta = DSDF_PRINCIPLED_ETA;
    const float ct = sqrtf(fmaxf(1.f - (1.f - c * c) / (eta * eta), 0.f));
    const float dct = c / (eta * eta * ct);
    const float den_s = c + eta * ct, den_p = ct + eta * c;
    const float a_s = (c - eta * ct) / den_s, a_p = (ct - eta * c) / den_p;
    const float F = 0.5f * (a_s * a_s + a_p * a_p);
    const float k = 2.f * eta * (ct - c * dct);
    const float dF = a_s * k / (den_s * den_s) - a_p * k / (den_p * den_p);
    T.ks = F * D * Gx * Gy / (4.f * x);
    const float dc_du = 0.25f / c, dch_du = -ch * 0.25f / (c * c), dch_dxy = 0.5f / c;
    T.dks[0] = T.ks * (dlnGx_dx - 1.f / x + dlnD_dch * dch_dxy);
    T.dks[1] = T.ks * (dlnGy_dy + dlnD_dch * dch_dxy);
    T.dks[2] = D * Gx * Gy / (4.f * x) * dF * dc_du + T.ks * dlnD_dch * dch_du;
    T.dks[3] = T.ks * (dlnD_da2 + dlnGx_da2 + dlnGy_da2) * da2_dr;
    return T;
}

#if DSDF_XF
// ---------------------------------------------------------------------------------------------------------------------------
// `use_mis` with the principled BSDF (sdf_direct_reparam.py:77-105 over the principled-* configs) -- only in the extended build
// (-DDSDF_XF=1, lib/variants/libdsdf_xf.so).  Principled::pdf and Principled::sample at the plugin defaults, restated from the
// published plugin like eval above [third-party, parity unpinned]: main_specular_sampling_rate = diffuse_reflectance_sampling_rate
// = 1 and no transmission / clearcoat leave two lobes chosen with probability 1/2 each --
//   sample1 <  1/2 : wo = square_to_cosine_hemisphere(sample2)
//   sample1 >= 1/2 : wo = reflect(wi, m), m a VISIBLE GGX normal (MicrofacetDistribution(GGX, alpha, sample_visible).sample:
//                    stretch wi, sample_visible_11 in its projected-disk form, rotate, unstretch, normalise) from the SAME sample2
//   pdf(wo) = 1/2 D(h) G1(wi, h) |wi.h| / |cos_i| / (4 |wo.h|)  +  1/2 cos_o / pi      (both directions on the front side)
// The oracle checks that the two are consistent (the pdf integrates to the accepted fraction of the samples, E[cos / pdf] = pi).
// ---------------------------------------------------------------------------------------------------------------------------
DSDF_HD float principled_pdf(float x, float y, float u, float r) {
    const float inv_pi = 0.3183098861837907f;
    const float alpha = fmaxf(r * r, 0.001f), a2 = alpha * alpha;
    const float c = sqrtf(fmaxf(0.5f * (1.f + u), 1e-12f));             // wi . h = wo . h
    const float ch = (x + y) / (2.f * c);                                // n . h
    const float s = 1.f - ch * ch + a2 * ch * ch;
    const float D = a2 * inv_pi / (s * s);
    const float qx = sqrtf(1.f + a2 * (1.f / (x * x) - 1.f));
    const float Gx = 2.f / (1.f + qx);
    return 0.5f * D * Gx / (4.f * x) + 0.5f * y * inv_pi;
}

// mitsuba warp.h square_to_uniform_disk_concentric
DSDF_HD void concentric_disk(float u0, float u1, float &dx, float &dy) {
    const float x = 2.f * u0 - 1.f, y = 2.f * u1 - 1.f;
    const bool zero = x == 0.f && y == 0.f, q13 = fabsf(x) < fabsf(y);
    const float rr = q13 ? y : x, rp = q13 ? x : y;
    float phi = zero ? 0.f : 0.7853981633974483f * rp / rr;
    if (q13) phi = 1.5707963267948966f - phi;
    if (zero) phi = 0.f;
    dx = rr * cosf(phi); dy = rr * sinf(phi);
}

// Principled::sample in the local shading frame (wi.z = cos_i > 0).  Returns false when the sample is rejected.
DSDF_HD bool principled_sample(float wix, float wiy, float wiz, float r, float u1, float ua, float ub, float &wox, float &woy, float &woz) {
    if (!(wiz > 0.f)) return false;
    float dx, dy;
    concentric_disk(ua, ub, dx, dy);
    if (u1 < 0.5f) {
        wox = dx; woy = dy; woz = sqrtf(fmaxf(1.f - dx * dx - dy * dy, 0.f));
        return woz > 0.f;
    }
    const float alpha = fmaxf(r * r, 0.001f);
    float px = alpha * wix, py = alpha * wiy, pz = wiz;                  // step 1: stretch wi
    const float inl = 1.f / sqrtf(px * px + py * py + pz * pz);
    px *= inl; py *= inl; pz *= inl;
    const float st2 = 1.f - pz * pz;
    float cos_phi = 1.f, sin_phi = 0.f;
    if (fabsf(st2) > 4.f * 1.1920929e-7f) {
        const float is = 1.f / sqrtf(st2);
        cos_phi = fminf(fmaxf(px * is, -1.f), 1.f); sin_phi = fminf(fmaxf(py * is, -1.f), 1.f);
    }
    // step 2: sample_visible_11(cos_theta = pz, sample)
    const float sfac = 0.5f * (1.f + pz);
    const float qy = sqrtf(fmaxf(1.f - dx * dx, 0.f)) * (1.f - sfac) + dy * sfac;
    const float qz = sqrtf(fmaxf(1.f - dx * dx - qy * qy, 0.f));
    const float sin_t = sqrtf(fmaxf(1.f - pz * pz, 0.f));
    const float norm = 1.f / (sin_t * qy + pz * qz);
    const float slx = (pz * qy - sin_t * qz) * norm, sly = dx * norm;
    // step 3: rotate & unstretch; step 4: normal
    const float sx = (cos_phi * slx - sin_phi * sly) * alpha, sy = (sin_phi * slx + cos_phi * sly) * alpha;
    const float iml = 1.f / sqrtf(sx * sx + sy * sy + 1.f);
    const float mx = -sx * iml, my = -sy * iml, mz = iml;
    const float wim = wix * mx + wiy * my + wiz * mz;
    wox = 2.f * wim * mx - wix; woy = 2.f * wim * my - wiy; woz = 2.f * wim * mz - wiz;
    const float wom = wox * mx + woy * my + woz * mz;
    return woz > 0.f && wim > 0.f && wom > 0.f;                          // reflect && mac_mic_compatibility
}
#endif
