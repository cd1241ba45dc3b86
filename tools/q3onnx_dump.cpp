// q3onnx_dump -- prints what the engine's ONNX reader sees in a model file: I/O, initialisers, the op histogram with the HIP kernel that
// serves each op, and whether the graph carries the streaming-decoder contract of /root/reference/src/models/onnx.rs:355-455.
//   q3onnx_dump model.onnx [--nodes]
#include "q3tts.h"
#include <cstdio>
#include <cstring>
#include <vector>
int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: q3onnx_dump model.onnx [--nodes]\n"); return 2; }
    q3tts_onnx* m = nullptr;
    if (q3tts_onnx_open(argv[1], &m) != Q3TTS_OK) { fprintf(stderr, "%s\n", q3tts_last_error()); return 1; }
    const int64_t n = q3tts_onnx_summary(m, nullptr, 0);
    std::vector<char> buf((size_t)n);
    q3tts_onnx_summary(m, buf.data(), n);
    fputs(buf.data(), stdout);
    if (argc > 2 && !strcmp(argv[2], "--nodes")) {
        int32_t nn = 0;
        q3tts_onnx_counts(m, &nn, nullptr, nullptr, nullptr);
        for (int i = 0; i < nn; i++) {
            const char *op, *name; int32_t ni, no, na;
            q3tts_onnx_node(m, i, &op, &name, &ni, &no, &na);
            printf("%4d %-24s %-28s", i, op, name);
            for (int j = 0; j < ni; j++) printf(" %s", q3tts_onnx_node_input(m, i, j));
            printf(" ->");
            for (int j = 0; j < no; j++) printf(" %s", q3tts_onnx_node_output(m, i, j));
            printf("\n");
        }
    }
    q3tts_onnx_close(m);
    return 0;
}
