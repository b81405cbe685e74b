// A minimal libHM client, written against the libHMDecoder interface only (the loop documented in libHMDecoder.h:36-77): reads an
// Annex B file, pushes NAL units, drains pictures, prints "POC width height [first luma sample]" per output picture.
// Built by tests/test_facade_client.py twice: against include/hmdec.h and -- where the reference checkout exists -- against
// libHM's own libHMDecoder.h, to show that libhmdec.so is a drop-in at source level.
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifdef USE_REFERENCE_HEADER
#include "libHMDecoder.h"
extern "C" void hmdec_set_parse_only(libHMDec_context* ctx, int on);      // the one call a GPU-less host needs
#else
#include "hmdec.h"
#endif

static std::vector<std::vector<unsigned char>> split(const std::vector<unsigned char>& b) {
  std::vector<size_t> starts;
  for (size_t i = 0; i + 2 < b.size(); i++) if (b[i] == 0 && b[i + 1] == 0 && b[i + 2] == 1) { starts.push_back(i + 3); i += 2; }
  std::vector<std::vector<unsigned char>> out;
  for (size_t k = 0; k < starts.size(); k++) {
    size_t e = k + 1 < starts.size() ? starts[k + 1] - 3 : b.size();
    while (e > starts[k] && b[e - 1] == 0) e--;
    out.emplace_back(b.begin() + starts[k], b.begin() + e);
  }
  return out;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<unsigned char> data;
  unsigned char buf[4096];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) data.insert(data.end(), buf, buf + n);
  fclose(f);
  libHMDec_context* dec = libHMDec_new_decoder();
  if (!dec) return 3;
  const bool parse_only = argc > 2;
  if (parse_only) hmdec_set_parse_only(dec, 1);
  printf("version %s\n", libHMDec_get_version());
  const bool keep_going = getenv("HMDEC_CLIENT_KEEP_GOING") != nullptr;
  int errors = 0;
  const auto nals = split(data);
  for (size_t i = 0; i < nals.size(); i++) {
    const bool eof = i + 1 == nals.size();
    bool again = true;
    while (again) {
      bool new_picture = false, check_output = false;
      if (libHMDec_push_nal_unit(dec, nals[i].data(), (int)nals[i].size(), eof, new_picture, check_output) != LIBHMDEC_OK) {
        if (!keep_going) return 4;
        errors++;                      // robustness runs: drop the unit, go on with the next one
        new_picture = false;
      }
      if (check_output)
        while (libHMDec_picture* pic = libHMDec_get_picture(dec)) {
          short* y = libHMDEC_get_image_plane(pic, LIBHMDEC_LUMA);
          printf("POC %d %dx%d chroma %dx%d format %d depth %d", libHMDEC_get_POC(pic), libHMDEC_get_picture_width(pic, LIBHMDEC_LUMA),
                 libHMDEC_get_picture_height(pic, LIBHMDEC_LUMA), libHMDEC_get_picture_width(pic, LIBHMDEC_CHROMA_U),
                 libHMDEC_get_picture_height(pic, LIBHMDEC_CHROMA_V), (int)libHMDEC_get_chroma_format(pic), libHMDEC_get_internal_bit_depth(LIBHMDEC_LUMA));
          if (y) printf(" first %d", y[0]);
          std::vector<libHMDec_BlockValue>* v = libHMDEC_get_internal_info(dec, pic, LIBHMDEC_CU_PREDICTION_MODE);
          printf(" cus %zu\n", v ? v->size() : (size_t)0);
        }
      again = new_picture;
    }
  }
  libHMDEC_clear_internal_info(dec);
  if (libHMDec_free_decoder(dec) != LIBHMDEC_OK) return 5;
  return errors ? 4 : 0;
}
