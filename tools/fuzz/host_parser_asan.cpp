// tools/fuzz/host_parser_asan.cpp -- the host-side marker walk (Identify + Decode with a recording scan handler) over files
// given on the command line, built with AddressSanitizer / UBSan on the CPU (no GPU needed: nothing here touches HIP).
//   g++ -std=c++17 -g -O1 -fsanitize=address,undefined -I jpeglibrary_amd/csrc tools/fuzz/host_parser_asan.cpp \
//       jpeglibrary_amd/csrc/host_parser.cpp -o /tmp/host_asan && /tmp/host_asan files...
#include <cstdio>
#include <vector>

#include "../../include/jpgpu.h"
#include "host.h"

using namespace jpgpu;

namespace {
class Recorder final : public ScanHandler {
  public:
    void on_frame(HostDecoder &dec, int sof) override {
        baseline_ = false;
        flush();
        if (sof == kSOF0 || sof == kSOF1) {
            geo_ = BaselineGeometry::latch(dec, dec.frame_header());
            baseline_ = true;
        } else if (sof == kSOF2) {
            prog_.begin(dec, dec.frame_header());
        }
    }
    void on_scan(HostDecoder &dec, MarkerReader &reader, const ScanHeader &scan) override {
        const uint8_t *entropy = reader.remaining_bytes();
        const size_t len = (size_t)reader.remaining_byte_count();
        if (prog_.active()) {
            prog_.add_scan(dec, scan, entropy, len);
            return;
        }
        if (!baseline_) throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "unsupported frame", kDetailUnsupportedFrame);
        if (scan.num_components != 0) jobs_.push_back(make_scan_job(dec, geo_, scan, entropy, len, false));
        reader.try_advance((int)find_scan_end(entropy, len));
    }
    void on_dispose(HostDecoder &) override { flush(); }
    size_t jobs() const { return jobs_.size(); }

  private:
    void flush() {
        if (!prog_.active()) return;
        if (!prog_.scans().empty()) {
            jobs_.push_back(prog_.make_frame_job());
            for (ScanJob &j : prog_.scans()) jobs_.push_back(std::move(j));
        }
        prog_.reset();
    }
    std::vector<ScanJob> jobs_;
    BaselineGeometry geo_;
    ProgressiveFrame prog_;
    bool baseline_ = false;
};
}  // namespace

int main(int argc, char **argv) {
    size_t ok = 0, failed = 0, jobs = 0;
    for (int a = 1; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) continue;
        std::vector<uint8_t> d;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
        fclose(f);
        // the batch ingest's header-only walk: Identify up to the first SOS header, then Decode's loop with a handler that
        // plans the scan and skips to the end of the data (DeviceBatch::plan_file_headers)
        try {
            HostDecoder dec;
            dec.set_input(d.data(), d.size());
            size_t pos = 0;
            if (dec.identify_until_scan(false, &pos) && pos > d.size()) return 1;
            ok++;
        } catch (const DecodeError &) {
            failed++;
        }
        for (int optimizer_rules = 0; optimizer_rules < 2; optimizer_rules++) {
            try {
                HostDecoder dec;
                dec.set_input(d.data(), d.size());
                if (!optimizer_rules) dec.identify(false);
                Recorder rec;
                dec.decode(rec, true);
                jobs += rec.jobs();
                ok++;
            } catch (const DecodeError &) {
                failed++;
            }
        }
    }
    printf("walks ok %zu, refused %zu, scan jobs %zu\n", ok, failed, jobs);
    return 0;
}
