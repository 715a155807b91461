// ref_shim_features.cpp — caller shim around the REFERENCE's own SIFT (features/sift/SIFT_Anatomy_Image_Describer.hpp, header-only)
// for the real-image fixture of SURVEY.md 8(c): descriptors of the two SceauxCastle JPGs that ship with openMVG
// (openMVG_Samples/imageData/SceauxCastle/100_710{1,2}.jpg). TEST INFRASTRUCTURE ONLY: it produces tests/golden/sceaux_sift.npz
// (tests/golden/make_sceaux_golden.py), nothing in the product calls it.
//
// The reference's image/image_io.cpp cannot be compiled here without libpng / libtiff build trees, so the one function of it
// this path needs - ReadImage(path, bytes, w, h, depth), image_io.cpp:60-100 -> ReadJpg :102-150 - is provided below for JPEG
// files only, on the reference's vendored libjpeg (third_party/jpeg) with the same decompression calls as ReadJpgStream
// (:131-170). Everything downstream is reference code: the RGB -> gray conversion of image_io.hpp:360-395 and the describer.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <csetjmp>
#include <memory>
#include <vector>

extern "C" {
#include "jpeglib.h"
}

#include "openMVG/features/sift/SIFT_Anatomy_Image_Describer.hpp"
#include "openMVG/image/image_io.hpp"

namespace openMVG {
namespace image {

// the explicit instantiations image_io.hpp:23-27 declares extern (defined in image_io.cpp:25-29, which is not compiled here)
template class Image<unsigned char>;
template class Image<float>;
template class Image<double>;
template class Image<RGBColor>;
template class Image<RGBAColor>;

namespace {
struct ErrorJump { jpeg_error_mgr pub; jmp_buf jump; };
void on_error(j_common_ptr cinfo) { longjmp(reinterpret_cast<ErrorJump*>(cinfo->err)->jump, 1); }
}  // namespace

int ReadImage(const char* filename, std::vector<unsigned char>* ptr, int* w, int* h, int* depth) {
  FILE* file = fopen(filename, "rb");
  if (!file) return 0;
  jpeg_decompress_struct cinfo;
  ErrorJump err;
  cinfo.err = jpeg_std_error(&err.pub);
  err.pub.error_exit = &on_error;
  if (setjmp(err.jump)) { jpeg_destroy_decompress(&cinfo); fclose(file); return 0; }
  jpeg_create_decompress(&cinfo);
  jpeg_stdio_src(&cinfo, file);
  jpeg_read_header(&cinfo, TRUE);
  jpeg_start_decompress(&cinfo);
  *w = cinfo.output_width; *h = cinfo.output_height; *depth = cinfo.output_components;
  const size_t stride = (size_t)cinfo.output_width * cinfo.output_components;
  ptr->resize(stride * cinfo.output_height);
  while (cinfo.output_scanline < cinfo.output_height) {
    JSAMPROW row[1] = {ptr->data() + stride * cinfo.output_scanline};
    jpeg_read_scanlines(&cinfo, row, 1);
  }
  jpeg_finish_decompress(&cinfo);
  jpeg_destroy_decompress(&cinfo);
  fclose(file);
  return 1;
}

}  // namespace image
}  // namespace openMVG

namespace {
std::unique_ptr<openMVG::features::SIFT_Regions> g_regions;
int g_w = 0, g_h = 0;
}

extern "C" {

// Describes a JPEG with the reference's SIFT (default Params: first octave 0, 6 octaves, 3 scales, RootSIFT; NORMAL preset).
// Returns the number of regions (< 0: the image could not be read).
int ref_sift_describe_jpeg(const char* path) {
  openMVG::image::Image<unsigned char> gray;
  if (!openMVG::image::ReadImage(path, &gray)) return -1;
  g_w = gray.Width(); g_h = gray.Height();
  openMVG::features::SIFT_Anatomy_Image_describer describer;
  g_regions = describer.Describe_SIFT_Anatomy(gray, nullptr);
  return g_regions ? (int)g_regions->RegionCount() : -2;
}

void ref_sift_image_size(int* w, int* h) { *w = g_w; *h = g_h; }

// desc: n x 128 bytes (Regions::DescriptorRawData, scalar_regions.hpp:93); feat: n x 4 floats (x, y, scale, orientation)
void ref_sift_copy(uint8_t* desc, float* feat) {
  if (!g_regions) return;
  const size_t n = g_regions->RegionCount();
  if (n) memcpy(desc, g_regions->DescriptorRawData(), n * 128);
  for (size_t k = 0; k < n; ++k) {
    const auto& f = g_regions->Features()[k];
    feat[4 * k] = f.x(); feat[4 * k + 1] = f.y(); feat[4 * k + 2] = f.scale(); feat[4 * k + 3] = f.orientation();
  }
}

}  // extern "C"
