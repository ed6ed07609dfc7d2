// Test helper (CPU only): writes the float32 columns of a raw file as a binary_compressed PCD with the host programs' writer,
// and round-trips a byte file through lzf_compress / lzf_decompress.
//   pcd_compressed_check pcd <raw_f32> <n_points> <out.pcd> <field> [<field> ...]
//   pcd_compressed_check lzf <in_bytes> <out_bytes>      (prints the compressed size)
//   pcd_compressed_check read <in.pcd> <out_raw_f32> <field> [<field> ...]   (the programs' PCD reader; prints the point count)
//   pcd_compressed_check png <in.png> <out_raw_u16>      (the programs' depth PNG reader; prints "w h")
//   pcd_compressed_check pngw <in_raw_u16> <w> <h> <out.png>
//   pcd_compressed_check log <in.log> <out.log>          (load_log -> save_log; prints the entry count)
//   pcd_compressed_check ctr <in.ctr> <num> <res> <out_raw_f32>
//   pcd_compressed_check camera <in.txt> <out_raw_f32>   (6 floats)
#include "../../elasticreconstruction_amd/csrc/host/er_formats.h"

int main(int argc, char** argv) {
  if (argc >= 6 && std::string(argv[1]) == "pcd") {
    const size_t n = (size_t)atol(argv[3]);
    std::vector<std::string> names(argv + 5, argv + argc);
    std::vector<float> raw(names.size() * n);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(raw.data(), 4, raw.size(), f) != raw.size()) return 2;
    fclose(f);
    std::vector<const float*> cols;
    for (size_t c = 0; c < names.size(); c++) cols.push_back(raw.data() + c * n);
    return erfmt::save_pcd_compressed(argv[4], names, cols, n) ? 0 : 3;
  }
  if (argc >= 5 && std::string(argv[1]) == "read") {
    std::vector<std::string> names(argv + 4, argv + argc);
    std::vector<std::vector<float>> cols;
    size_t n = 0;
    if (!erfmt::load_pcd_fields(argv[2], names, cols, n)) return 5;
    FILE* f = fopen(argv[3], "wb");
    for (auto& c : cols)
      if (n) fwrite(c.data(), 4, n, f);
    fclose(f);
    printf("%zu\n", n);
    return 0;
  }
  if (argc == 4 && std::string(argv[1]) == "png") {
    int w = 0, h = 0;
    std::vector<uint16_t> px;
    if (!erfmt::load_png16(argv[2], w, h, px)) return 6;
    FILE* f = fopen(argv[3], "wb");
    fwrite(px.data(), 2, px.size(), f);
    fclose(f);
    printf("%d %d\n", w, h);
    return 0;
  }
  if (argc == 6 && std::string(argv[1]) == "pngw") {
    const int w = atoi(argv[3]), h = atoi(argv[4]);
    std::vector<uint16_t> px((size_t)w * h);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(px.data(), 2, px.size(), f) != px.size()) return 2;
    fclose(f);
    return erfmt::save_png16(argv[5], w, h, px.data()) ? 0 : 7;
  }
  if (argc == 4 && std::string(argv[1]) == "log") {
    std::vector<erfmt::FramedTransformation> v;
    if (!erfmt::load_log(argv[2], v) || !erfmt::save_log(argv[3], v)) return 8;
    printf("%zu\n", v.size());
    return 0;
  }
  if (argc == 6 && std::string(argv[1]) == "ctr") {
    std::vector<float> g;
    if (!erfmt::load_ctr(argv[2], atoi(argv[3]), atoi(argv[4]), g)) return 9;
    FILE* f = fopen(argv[5], "wb");
    fwrite(g.data(), 4, g.size(), f);
    fclose(f);
    return 0;
  }
  if (argc == 4 && std::string(argv[1]) == "camera") {
    float cam[6];
    erfmt::load_camera(argv[2], cam);
    FILE* f = fopen(argv[3], "wb");
    fwrite(cam, 4, 6, f);
    fclose(f);
    return 0;
  }
  if (argc == 4 && std::string(argv[1]) == "lzf") {
    FILE* f = fopen(argv[2], "rb");
    if (!f) return 2;
    std::vector<uint8_t> in, packed, back;
    uint8_t buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) in.insert(in.end(), buf, buf + got);
    fclose(f);
    erfmt::lzf_compress(in.data(), in.size(), packed);
    if (!erfmt::lzf_decompress(packed.data(), packed.size(), back, in.size())) return 4;
    f = fopen(argv[3], "wb");
    if (!back.empty()) fwrite(back.data(), 1, back.size(), f);
    fclose(f);
    printf("%zu\n", packed.size());
    return 0;
  }
  return 1;
}
