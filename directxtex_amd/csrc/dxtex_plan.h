// Host-side resolution of ConvertScanline (DirectXTexConvert.cpp:3080-3854) for the non-depth formats this
// library handles: which of the per-texel steps (sRGB decode, range conversion, channel shuffle, sRGB encode) a
// given (input format, output format, TEX_FILTER flags) triple needs. The kernels then run apply_plan() per texel.
#pragma once
#include "dxtex_formats.h"
#include "dxtex_store.h"

namespace dxtex
{
enum : uint32_t
{
    TF_FLOAT_X2BIAS = 0x200, TF_COPY_RED = 0x1000, TF_COPY_GREEN = 0x2000, TF_COPY_BLUE = 0x4000, TF_COPY_ALPHA = 0x8000,
    TF_SRGB_IN = 0x1000000, TF_SRGB_OUT = 0x2000000,
};

// The depth branch of ConvertScanline (:3186-3434) as a TDP word (dxtex_device.h); in / out differ in FC_DEPTH
inline int resolve_depth_steps(const FmtInfo& in, const FmtInfo& out, uint32_t flags)
{
    int a = 0, b = 0, c = 0;
    if (in.cls & FC_DEPTH)
    {
        if (in.cls & FC_STENCIL) a = (out.cls & FC_UNORM) ? TDP_S2A_UNORM : (out.cls & FC_SNORM) ? TDP_S2A_SNORM : TDP_S2A_RAW;
        if ((out.cls & FC_UNORM) && (in.cls & FC_FLOAT)) b = TDP_D2RGB_SAT;
        else if (out.cls & FC_SNORM) b = (in.cls & FC_UNORM) ? TDP_D2RGB_U2S : TDP_D2RGB_CLAMPS;
        else b = TDP_D2RGB_RAW;
    }
    else
    {
        switch (flags & (TF_COPY_RED | TF_COPY_GREEN | TF_COPY_BLUE | TF_COPY_ALPHA))
        {
        case TF_COPY_GREEN: b = TDP_X_FROM_Y; break;
        case TF_COPY_BLUE: b = TDP_X_FROM_Z; break;
        case TF_COPY_ALPHA: b = TDP_X_FROM_W; break;
        case TF_COPY_RED: break;
        default: if ((in.cls & FC_UNORM) && (in.cls & (FC_R | FC_G | FC_B)) == (FC_R | FC_G | FC_B)) b = TDP_X_GRAY; break;
        }
        if (out.cls & FC_UNORM) c = (in.cls & FC_SNORM) ? TDP_X_S2U : (in.cls & FC_FLOAT) ? TDP_X_SAT : 0;
        if (out.cls & FC_STENCIL) a = (in.cls & FC_UNORM) ? TDP_A2S_UNORM : (in.cls & FC_SNORM) ? TDP_A2S_SNORM : TDP_A2S_RAW;
    }
    return a | (b << 4) | (c << 8);
}

inline ConvertPlan resolve_convert_plan(const FmtInfo& in, const FmtInfo& out, uint32_t flags)
{
    ConvertPlan p; p.srgbIn = 0; p.tcv = TCV_NONE; p.tsw = TSW_NONE; p.srgbOut = 0; p.depth = 0;

    // :3123-3167
    if (in.cls & FC_SRGB) flags |= TF_SRGB_IN;
    if (in.format == FMT_A8_UNORM || in.format == FMT_R10G10B10_XR_BIAS_A2_UNORM) flags &= ~TF_SRGB_IN;       // :3136-3139
    if (out.cls & FC_SRGB) flags |= TF_SRGB_OUT;
    if (out.format == FMT_A8_UNORM || out.format == FMT_R10G10B10_XR_BIAS_A2_UNORM) flags &= ~TF_SRGB_OUT;    // :3156-3159
    if ((flags & (TF_SRGB_IN | TF_SRGB_OUT)) == (TF_SRGB_IN | TF_SRGB_OUT)) flags &= ~(TF_SRGB_IN | TF_SRGB_OUT);
    if ((flags & TF_SRGB_IN) && !(in.cls & FC_DEPTH) && (in.cls & (FC_FLOAT | FC_UNORM))) p.srgbIn = 1;        // :3170-3180
    if ((flags & TF_SRGB_OUT) && !(out.cls & FC_DEPTH) && (out.cls & (FC_FLOAT | FC_UNORM))) p.srgbOut = 1;    // :3843-3853

    // the reference compares its CONVF_* words; what can differ among our formats: type class, channel set, BC-ness,
    // BGR order. BGR-only differences reach no branch below, so they can be left out of the test.
    const uint32_t kDiffMask = FC_UNORM | FC_SNORM | FC_FLOAT | FC_BC | FC_R | FC_G | FC_B | FC_A | FC_POS_ONLY | FC_UINT | FC_SINT | FC_XR | FC_YUV | FC_DEPTH | FC_STENCIL | FC_PACKED;
    const uint32_t diff = (in.cls ^ out.cls) & kDiffMask;
    if (!diff) return p;

    const bool x2 = (flags & TF_FLOAT_X2BIAS) != 0;
    if (diff & FC_DEPTH) p.depth = resolve_depth_steps(in, out, flags);                                // :3186-3434
    else if (out.cls & FC_DEPTH) { if ((diff & FC_FLOAT) && (in.cls & FC_FLOAT)) p.depth = TDP_X_SAT << 8; }   // depth -> depth, :3435-3451
    else if (out.cls & FC_UNORM)
    {
        if (in.cls & FC_SNORM) p.tcv = TCV_SNORM_TO_UNORM;                                            // :3457-3463
        else if (in.cls & FC_FLOAT) p.tcv = (!(in.cls & FC_POS_ONLY) && x2) ? TCV_X2BIAS_TO_UNORM : TCV_SATURATE;   // :3465-3489
    }
    else if (out.cls & FC_SNORM)
    {
        if (in.cls & FC_UNORM) p.tcv = TCV_UNORM_TO_SNORM;                                            // :3495-3501
        else if (in.cls & FC_FLOAT) p.tcv = ((in.cls & FC_POS_ONLY) && x2) ? TCV_SAT_TO_SNORM : TCV_CLAMP_SNORM;     // :3503-3527
    }
    else if (diff & FC_UNORM)
    {
        if ((out.cls & FC_FLOAT) && !(out.cls & FC_POS_ONLY) && x2) p.tcv = TCV_UNORM_TO_SNORM;        // UNORM (x2 bias) -> FLOAT, :3529-3543
    }
    else if ((diff & FC_POS_ONLY) && x2)
    {
        // :3545-3587
        if (in.cls & FC_POS_ONLY) { if (out.cls & FC_FLOAT) p.tcv = TCV_SAT_TO_SNORM; }               // FLOAT (positive only, x2 bias) -> FLOAT
        else if (out.cls & FC_POS_ONLY)
        {
            if (in.cls & FC_FLOAT) p.tcv = TCV_X2BIAS_TO_UNORM;                                       // FLOAT -> FLOAT (positive only, x2 bias)
            else if (in.cls & FC_SNORM) p.tcv = TCV_SNORM_TO_UNORM;                                   // SNORM -> FLOAT (positive only, x2 bias)
        }
    }

    const uint32_t inRGBA = in.cls & (FC_R | FC_G | FC_B | FC_A), outRGBA = out.cls & (FC_R | FC_G | FC_B | FC_A);
    const uint32_t inRGB = in.cls & (FC_R | FC_G | FC_B), outRGB = out.cls & (FC_R | FC_G | FC_B);
    const uint32_t kRGB = FC_R | FC_G | FC_B;
    if (outRGBA == FC_A && !(in.cls & FC_A))
    {
        // !A -> A format (:3596-3652)
        switch (flags & (TF_COPY_RED | TF_COPY_GREEN | TF_COPY_BLUE))
        {
        case TF_COPY_GREEN: p.tsw = TSW_SPLAT_Y; break;
        case TF_COPY_BLUE: p.tsw = TSW_SPLAT_Z; break;
        case TF_COPY_RED: p.tsw = TSW_SPLAT_X; break;
        default: p.tsw = ((in.cls & FC_UNORM) && inRGB == kRGB) ? TSW_GRAY_SPLAT : TSW_SPLAT_X; break;
        }
    }
    else if (inRGBA == FC_A && !(out.cls & FC_A)) p.tsw = TSW_A_TO_RGB;                               // :3654-3664
    else if (inRGB == FC_R)
    {
        if (outRGB == kRGB) p.tsw = TSW_R_TO_RGB;                                                      // :3667-3679
        else if (outRGB == (FC_R | FC_G)) p.tsw = TSW_R_TO_RG;                                         // :3680-3691
    }
    else if (inRGB == kRGB)
    {
        if (outRGB == FC_R)
        {
            // RGB(A) -> R format (:3696-3771)
            switch (flags & (TF_COPY_RED | TF_COPY_GREEN | TF_COPY_BLUE | TF_COPY_ALPHA))
            {
            case TF_COPY_GREEN: p.tsw = TSW_G_TO_R; break;
            case TF_COPY_BLUE: p.tsw = TSW_B_TO_R; break;
            case TF_COPY_ALPHA: p.tsw = TSW_A_TO_R; break;
            case TF_COPY_RED: break;
            default: if (in.cls & FC_UNORM) p.tsw = TSW_RGB_TO_R_GRAY; break;
            }
        }
        else if (outRGB == (FC_R | FC_G))
        {
            // RGB(A) -> RG format (:3773-3838)
            if ((flags & TF_COPY_ALPHA) && (in.cls & FC_A))
            {
                switch (flags & (TF_COPY_RED | TF_COPY_GREEN | TF_COPY_BLUE | TF_COPY_ALPHA))
                {
                case TF_COPY_GREEN | TF_COPY_ALPHA: p.tsw = TSW_GA_TO_RG; break;
                case TF_COPY_BLUE | TF_COPY_ALPHA: p.tsw = TSW_BA_TO_RG; break;
                default: p.tsw = TSW_RA_TO_RG; break;
                }
            }
            else
            {
                switch (flags & (TF_COPY_RED | TF_COPY_GREEN | TF_COPY_BLUE))
                {
                case TF_COPY_RED | TF_COPY_BLUE: p.tsw = TSW_RB_TO_RG; break;
                case TF_COPY_GREEN | TF_COPY_BLUE: p.tsw = TSW_GB_TO_RG; break;
                default: break;
                }
            }
        }
    }
    return p;
}
} // namespace dxtex
