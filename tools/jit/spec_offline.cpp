// tools/jit/spec_offline.cpp -- writes the piece_spec_config.h jit.hpp would generate for an aligner, so that
// piece_spec.hip can be compiled with hipcc offline (ISA inspection, resource counts; tools/jit/spec_offline.sh).
//   spec_offline <adapter> <max_error_rate> <flags> <min_overlap> <max_len> <ragged 0|1> <out_dir> [--rtc]
// --rtc: also run the very hiprtc path of the library (no GPU needed) and write the code object to <out_dir>/spec.hsaco.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "locate_fast.hpp"
#include "jit.hpp"

using namespace atr;

int main(int argc, char **argv) {
    if (argc < 8) { fprintf(stderr, "usage: %s adapter e flags min_overlap max_len ragged out_dir [--rtc]\n", argv[0]); return 2; }
    const char *ad = argv[1];
    atr_aligner *a = nullptr;
    if (aligner_create(ad, (int)strlen(ad), atof(argv[2]), atoi(argv[3]), 0, 0, atoi(argv[4]), 1, &a) != ATR_OK) { fprintf(stderr, "aligner_create failed\n"); return 1; }
    const int max_len = atoi(argv[5]);
    const bool ragged = atoi(argv[6]) != 0;
    const int nw = (max_len + 31) / 32, n = ragged ? 32 * nw : max_len;
    const FilterParams fp = filter_params(a->peq, a->codes, a->p.m, a->flags, false, a->p.thr, a->p.min_overlap, true);
    PieceParams pp;
    if (!piece_params(a->codes, a->p.m, fp.rows, a->p.k, a->flags, false, a->table_kind == ATR_TABLE_CUSTOM, fp.thr_row, n, pp, a->p.thr, a->p.min_overlap)) {
        fprintf(stderr, "outside the two-pass envelope\n");
        return 1;
    }
    const std::string cfg = jit::spec_config(a, fp, pp, n);
    const std::string dir = argv[7];
    FILE *f = fopen((dir + "/piece_spec_config.h").c_str(), "w");
    if (!f) { perror("open"); return 1; }
    fwrite(cfg.data(), 1, cfg.size(), f);
    fclose(f);
    printf("NW=%d RAGGED=%d n=%d blen=%d llen=%d tlen=%d steps=%d xlo=%d xhi=%d\n", nw, (int)ragged, n, pp.blen, pp.llen, pp.tlen, pp.steps, pp.xlo, pp.xhi);
    if (argc > 8 && !strcmp(argv[8], "--rtc")) {
        std::string log;
        const std::vector<char> code = jit::compile_spec(cfg, nw, ragged, 0, "gfx950", &log);
        if (code.empty()) { fprintf(stderr, "hiprtc failed:\n%s\n", log.c_str()); return 1; }
        f = fopen((dir + "/spec.hsaco").c_str(), "wb");
        fwrite(code.data(), 1, code.size(), f);
        fclose(f);
        printf("hiprtc: %zu bytes of code object\n", code.size());
    }
    return 0;
}
