// Geometry of one causal 3-D convolution launch (conv_igemm.hip, conv256s.hip); see conv_igemm.hip for the layouts.
#pragma once

struct ConvGeom {
    // output positions
    int To, Ho, Wo;
    // input addressing: frame pitch (in positions) and row pitch, origin of the un-padded image
    int T, Hp, Wp, ih0, iw0;
    int kT, kH, kW, sT, sH, sW;
    int t_off;            // ti = max(to*sT + kt - t_off, 0)
    int pad;              // spatial padding of the conv (0 or 1): hi = ho*sH + kh - pad
    // output addressing
    int oHp, oWp, oh0, ow0;
};
