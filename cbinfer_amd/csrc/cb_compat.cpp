// Reference-signature shims.  Built three times (see Makefile) into
//   compat/cbconv2d_cg_backend_<machine>.so       (-DCB_COMPAT_CG -DCB_COMPAT_DTYPE=CB_F32)
//   compat/cbconv2d_cg_half_backend_<machine>.so  (-DCB_COMPAT_CG -DCB_COMPAT_DTYPE=CB_F16)
//   compat/cbconv2d_fg_backend_<machine>.so       (-DCB_COMPAT_FG)
// each exporting exactly the symbols the reference's cffi cdef declares (conv2d_cg.py:6-38,
// conv2d_fg.py:14-29) and forwarding to libcbinfer_hip.so on the default stream.  The six launch
// geometry ints the reference computes in Python are accepted and ignored: geometry is chosen inside
// the library.  Like the reference launchers these return void; failures are reported on stderr.
#include <stdio.h>

#include "../../include/cbinfer_hip.h"

static void cb_report(const char* what, int status) {
    if (status != CB_OK) fprintf(stderr, "cbinfer compat: %s failed: %s\n", what, cbinfer_status_string(status));
}

extern "C" {

#ifdef CB_COMPAT_CG
// cbconv2d_cg_backend.cu:83-100 (half: cbconv2d_cg_half_backend.cu)
void changeDetection(int, int, int, int, int, int, const float* input, float* oldinput,
                     bool* changeMatrix, const int width, const int height, const int nInputPlane,
                     const int kHHalf, const int kWHalf, const float diffThreshold,
                     const bool updateInputState) {
    cb_report("changeDetection",
              cbinfer_change_detection(input, oldinput, (int8_t*)changeMatrix, width, height,
                                       nInputPlane, kHHalf, kWHalf, diffThreshold, updateInputState,
                                       CB_COMPAT_DTYPE, nullptr));
}

// cbconv2d_cg_backend.cu:126-136
void changePropagation(int, int, int, int, int, int, const bool* changeMatrixIn, bool* changeMatrixOut,
                       const int width, const int height, const int kHHalf, const int kWHalf) {
    cb_report("changePropagation",
              cbinfer_change_propagation((const int8_t*)changeMatrixIn, (int8_t*)changeMatrixOut, width,
                                         height, kHHalf, kWHalf, nullptr));
}

// cbconv2d_cg_backend.cu:163-173
void genXMatrix(int, int, int, int, int, int, float* columns, const float* input,
                const int* changeList, const int kW, const int kH, const int nInputPlane,
                const int width, const int height, const int numChanges) {
    cb_report("genXMatrix", cbinfer_gen_x_matrix(columns, input, changeList, kW, kH, nInputPlane, width,
                                                 height, numChanges, nullptr, CB_COMPAT_DTYPE, nullptr));
}

// cbconv2d_cg_backend.cu:191-197
void updateOutput(int, int, int, int, int, int, float* columnsOut, float* output, int* changeList,
                  int numOutputPixel, int numChanges, int nOutputPlane, bool relu) {
    cb_report("updateOutput",
              cbinfer_update_output(columnsOut, output, changeList, numOutputPixel, numChanges, nullptr,
                                    nOutputPlane, relu, CB_COMPAT_DTYPE, nullptr));
}

// cbconv2d_cg_backend.cu:229-240 (stride is fixed to 2x2 by the caller, conv2d_cg.py:71)
void maxPool2d(int, int, float* input, float* output, int* changeIndexes, int numChanges, int numCh,
               int iheight, int iwidth, int oheight, int owidth, int stridey, int stridex) {
    if (stridey != 2 || stridex != 2) {
        cb_report("maxPool2d (stride must be 2x2)", CB_ERR_UNSUPPORTED);
        return;
    }
    cb_report("maxPool2d", cbinfer_max_pool2d(input, output, changeIndexes, numChanges, nullptr, numCh,
                                              iheight, iwidth, oheight, owidth, CB_COMPAT_DTYPE, nullptr));
}
#endif  // CB_COMPAT_CG

#ifdef CB_COMPAT_FG
// cbconv2d_fg_backend.cu:25-35
void changeDetectionFG(const float* input, const float* prevInput, float* diffs, char* changeMap,
                       const int numVals, const float threshold) {
    cb_report("changeDetectionFG", cbinfer_change_detection_fg(input, prevInput, diffs, (int8_t*)changeMap,
                                                               numVals, threshold, 0, nullptr));
}

// cbconv2d_fg_backend.cu:68-79
void updateOutputFG(int, int, int, int, int, int, const float* diffs, const float* weight, float* output,
                    const long* changeCoords, const int numOut, const int numIn, const int height,
                    const int width, const int kH, const int kW, const int numChanges) {
    cb_report("updateOutputFG",
              cbinfer_update_output_fg(diffs, weight, output, (const int64_t*)changeCoords, numOut, numIn,
                                       height, width, kH, kW, numChanges, nullptr));
}

// cbconv2d_fg_backend.cu:81-112
void conv2d_fg_cpu(const float* input, const float* prevInput, float* output, const float* weight,
                   const float threshold, const int no, const int ni, const int h, const int w,
                   const int kh, const int kw) {
    cbinfer_conv2d_fg_cpu(input, prevInput, output, weight, threshold, no, ni, h, w, kh, kw);
}
#endif  // CB_COMPAT_FG

}  // extern "C"
