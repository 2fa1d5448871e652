// Host-side native code (parser, FASTQ reader) under AddressSanitizer + UBSan: a CPU-only build of the three host sources
// driven over the reference's test data and a few irregular files.  (GPU sanitizers are not available on the pool.)
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -pthread -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
//       tests/tools/asan_host.cpp mcaller_amd/csrc/mc_parse.cpp mcaller_amd/csrc/mc_fastq.cpp mcaller_amd/csrc/mc_common.cpp \
//       -L/opt/rocm/lib -lamdhip64 -ldl -Wl,-rpath,/opt/rocm/lib -o /tmp/asan_host && /tmp/asan_host <tsv> <fastq>
// (and the same with -fsanitize=thread: the kept worker threads of mc_parallel_for)
#include "../../include/mcaller_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/stat.h>

static void write_file(const char *path, const std::string &text) {
    FILE *f = fopen(path, "wb");
    fwrite(text.data(), 1, text.size(), f);
    fclose(f);
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: asan_host <eventalign.tsv> <reads.fastq>\n"); return 2; }
    struct stat st;
    stat(argv[1], &st);
    const char *contigs[] = {"ecoli", "ecoli_syn"};
    long rows = -1;
    for (int threads : {1, 2, 3, 7, 16}) {
        mc_parsed *p = nullptr;
        if (mc_parse_eventalign(argv[1], 0, st.st_size, contigs, 2, threads, &p) != 0) { fprintf(stderr, "parse: %s\n", mc_last_error()); return 1; }
        mc_table_view v;
        mc_parsed_view(p, &v);
        if (rows >= 0 && rows != v.n_rows) { fprintf(stderr, "row count depends on the thread count\n"); return 1; }
        rows = v.n_rows;
        long chk = 0;
        for (int64_t i = 0; i < v.n_rows; ++i) chk += v.pos[i] + v.event_model_e4[2 * i] + v.event_model_e4[2 * i + 1] + v.event_idx[i] + v.flags[i];
        for (int r = 0; r < v.n_reads; ++r) chk += (long)strlen(mc_parsed_read_name(p, r));
        printf("threads %d: %lld rows, %d segments, %d reads, checksum %ld, %lld rows of unknown contigs\n", threads,
               (long long)v.n_rows, v.n_seg, v.n_reads, chk, (long long)mc_parsed_n_unknown(p));
        mc_parsed_free(p);
    }
    for (int parts : {1, 2, 5}) {
        int64_t cuts[8];
        if (mc_eventalign_read_cuts(argv[1], parts, cuts) != 0) { fprintf(stderr, "cuts: %s\n", mc_last_error()); return 1; }
        for (int i = 0; i < parts; ++i) {
            mc_parsed *p = nullptr;
            if (mc_parse_eventalign_range(argv[1], cuts[i], cuts[i + 1], contigs, 2, 3, &p) != 0) { fprintf(stderr, "range: %s\n", mc_last_error()); return 1; }
            mc_parsed_free(p);
        }
    }
    // a start inside the file (the reference's startline - 500 rule) and a range that ends mid-line
    for (int64_t start : {int64_t(1), int64_t(777), int64_t(st.st_size / 2), int64_t(st.st_size - 3)}) {
        mc_parsed *p = nullptr;
        if (mc_parse_eventalign(argv[1], start, st.st_size - 100, contigs, 2, 4, &p) == 0) mc_parsed_free(p);
    }
    for (int threads : {1, 4}) {
        mc_fastq *f = nullptr;
        if (mc_fastq_read_quality(argv[2], threads, &f) != 0) { fprintf(stderr, "fastq: %s\n", mc_last_error()); return 1; }
        const char *pool; const int64_t *off; const double *mean;
        const int64_t n = mc_fastq_view(f, &pool, &off, &mean);
        printf("fastq threads %d: %lld records, first key %.*s mean %.6f\n", threads, (long long)n, (int)(off[1] - off[0] - 1), pool, mean[0]);
        mc_fastq_free(f);
    }
    // irregular FASTQ files: every one must end in an error code or a result, never in a bad access
    const char *cases[] = {"", "@", "@a", "@a\n", "@a\nAC", "@a\nAC\n+", "@a\nAC\n+\nII", "@a\nAC\n+\nI", "x\n", "\n\n\n@a\nAC\n+\nII\n\n",
                           "@a\r\nAC\r\n+\r\nII\r\n", "@a b c\nAC\n+\n@@\n@b\nA\n+\n@", "@:\nA\n+\nI\n", "@_\nA\n+\nI\n", "@a\n\n+\n\n"};
    for (const char *text : cases) {
        write_file("/tmp/asan_case.fastq", text);
        mc_fastq *f = nullptr;
        const int rc = mc_fastq_read_quality("/tmp/asan_case.fastq", 3, &f);
        if (rc == 0) mc_fastq_free(f);
    }
    // irregular eventalign text
    const char *tsvs[] = {"", "\n", "a\tb\n", "ecoli\t5\tAAAAAA\tr1\tt\t3\t1.5\t0\t0\tAAAAAA\t2.5\t0\t0", "ecoli\t5\tAAAAAA\tr1\tt\t3\tx\t0\t0\tAAAAAA\t2.5\t0\t0\n",
                          "ecoli\t99999999999\tAAAAAA\tr1\tt\t3\t1.5\t0\t0\tAAAAAA\t2.5\t0\t0\n", "\t\t\t\t\t\t\t\t\t\t\t\t\n"};
    for (const char *text : tsvs) {
        write_file("/tmp/asan_case.tsv", text);
        mc_parsed *p = nullptr;
        if (mc_parse_eventalign("/tmp/asan_case.tsv", 0, (int64_t)strlen(text), contigs, 2, 2, &p) == 0) mc_parsed_free(p);
    }
    printf("asan_host: done\n");
    return 0;
}
