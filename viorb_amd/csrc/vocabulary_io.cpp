// viorb_amd/csrc/vocabulary_io.cpp — the two on-disk formats of the ORB vocabulary into the flat tree of
// viorb_vocabulary_create. Host code only.
//   text    TemplatedVocabulary::loadFromTextFile / saveToTextFile  (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1351-1460):
//           first line "k L scoring weighting", then one line per node 1..N in id order: "parent isLeaf d0 .. d31 weight"
//           (FORB::toString / fromString, Thirdparty/DBoW2/DBoW2/FORB.cpp:105-135: the 32 descriptor bytes as decimal integers)
//   binary  loadFromBinaryFile / saveToBinaryFile (:1462-1533, written by tools/bin_vocabulary.cc): u32 nb_nodes (root included),
//           u32 size_node (= 41), i32 k, i32 L, i32 scoring, i32 weighting, then nb_nodes - 1 records
//           { i32 parent; u8 descriptor[32]; f32 weight; u8 is_leaf }
// Node ids are file order (root = 0), a parent's children keep file order (children.push_back), word ids count the leaves in file
// order — exactly what TemplatedVocabulary::transform walks.
// Deliberate deviations, both from reads past the end of the data that the reference performs: its text loader turns the empty
// line that follows the last record into an extra child of the root with an UNINITIALISED descriptor (the stringstream extractions
// fail, cv::Mat::create does not clear), and its binary loader re-reads the last record once more after the final successful read
// (while(!f.eof())), appending a duplicate of the last node to its parent (never selected: ties go to the first child). Both loaders
// here stop at the last complete record.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "viorb_common.h"

namespace {

struct FlatTree {
    int k = 0, L = 0, scoring = 0, weighting = 0;
    std::vector<int> parent;            // per node (root: -1)
    std::vector<uint8_t> desc;          // [n][32]
    std::vector<double> weight;
    std::vector<uint8_t> leaf;
};

template <class T> T* dup(const std::vector<T>& v) {
    T* p = (T*)malloc(sizeof(T) * (v.size() ? v.size() : 1));
    if (p && !v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
    return p;
}

int build(const FlatTree& T, viorb_vocabulary** out, viorb_vocabulary_flat* flat) {
    const int n = (int)T.parent.size();
    std::vector<int> child_start(n + 1, 0), child_ids, word_id(n, -1), fill(n, 0);
    for (int i = 1; i < n; i++) {
        if (T.parent[i] < 0 || T.parent[i] >= n || T.parent[i] == i) { viorb::set_error("vocabulary node %d has parent %d", i, T.parent[i]); return VIORB_ERR_INVALID_ARG; }
        child_start[T.parent[i] + 1]++;
    }
    for (int i = 0; i < n; i++) child_start[i + 1] += child_start[i];
    child_ids.resize(child_start[n]);
    for (int i = 1; i < n; i++) child_ids[child_start[T.parent[i]] + fill[T.parent[i]]++] = i;      // file order inside every parent
    int words = 0;
    for (int i = 1; i < n; i++) if (T.leaf[i]) word_id[i] = words++;
    // a node the file marks as inner but that has no children would be treated as a word by the descent: give it the reference's answer
    // (Node::isLeaf() = children.empty(), word_id of a never-assigned word = 0)
    for (int i = 1; i < n; i++) if (!T.leaf[i] && child_start[i + 1] == child_start[i]) word_id[i] = 0;
    if (word_id[0] < 0 && child_start[1] == child_start[0]) word_id[0] = 0;
    if (flat) {
        flat->n_nodes = n; flat->k = T.k; flat->L = T.L; flat->n_words = words;
        flat->child_start = dup(child_start); flat->child_ids = dup(child_ids); flat->word_id = dup(word_id); flat->desc = dup(T.desc); flat->weight = dup(T.weight);
        if (!flat->child_start || !flat->child_ids || !flat->word_id || !flat->desc || !flat->weight) { viorb_vocabulary_flat_free(flat); viorb::set_error("out of memory"); return VIORB_ERR_INVALID_ARG; }
        return VIORB_OK;
    }
    return viorb_vocabulary_create(n, T.L, child_start.data(), child_ids.data(), T.desc.data(), word_id.data(), T.weight.data(), out);
}

int read_text(const char* path, FlatTree& T);
int read_binary(const char* path, FlatTree& T);

} // namespace

extern "C" {

void viorb_vocabulary_flat_free(viorb_vocabulary_flat* f) {
    if (!f) return;
    free(f->child_start); free(f->child_ids); free(f->word_id); free(f->desc); free(f->weight);
    memset(f, 0, sizeof(*f));
}
int viorb_vocabulary_read_file(const char* path, int binary, viorb_vocabulary_flat* out) {
    VIORB_REQUIRE(path && out, "null argument");
    memset(out, 0, sizeof(*out));
    FlatTree T;
    const int rc = binary ? read_binary(path, T) : read_text(path, T);
    return rc != VIORB_OK ? rc : build(T, nullptr, out);
}
int viorb_vocabulary_load_text(const char* path, viorb_vocabulary** out) {
    VIORB_REQUIRE(path && out, "null argument");
    *out = nullptr;
    FlatTree T;
    const int rc = read_text(path, T);
    return rc != VIORB_OK ? rc : build(T, out, nullptr);
}
int viorb_vocabulary_load_binary(const char* path, viorb_vocabulary** out) {
    VIORB_REQUIRE(path && out, "null argument");
    *out = nullptr;
    FlatTree T;
    const int rc = read_binary(path, T);
    return rc != VIORB_OK ? rc : build(T, out, nullptr);
}
} // extern "C"

namespace {
int read_text(const char* path, FlatTree& T) {
    FILE* f = fopen(path, "rb");
    if (!f) { viorb::set_error("cannot open %s", path); return VIORB_ERR_INVALID_ARG; }
    std::string all;
    { char buf[1 << 16]; size_t r; while ((r = fread(buf, 1, sizeof(buf), f)) > 0) all.append(buf, r); }
    fclose(f);
    const char* p = all.c_str();
    char* e = nullptr;
    T.k = (int)strtol(p, &e, 10); p = e; T.L = (int)strtol(p, &e, 10); p = e; T.scoring = (int)strtol(p, &e, 10); p = e; T.weighting = (int)strtol(p, &e, 10); p = e;
    // "if(m_k<0 || m_k>20 || m_L<1 || m_L>10 || n1<0 || n1>5 || n2<0 || n2>3)" (:1373)
    if (T.k < 0 || T.k > 20 || T.L < 1 || T.L > 10 || T.scoring < 0 || T.scoring > 5 || T.weighting < 0 || T.weighting > 3) {
        viorb::set_error("%s is not a vocabulary text file (k %d, L %d)", path, T.k, T.L); return VIORB_ERR_INVALID_ARG;
    }
    T.parent.push_back(-1); T.desc.assign(32, 0); T.weight.push_back(0.0); T.leaf.push_back(0);
    while (true) {
        const long pid = strtol(p, &e, 10);
        if (e == p) break;                                  // no further record
        p = e;
        const long is_leaf = strtol(p, &e, 10);
        if (e == p) break;
        p = e;
        uint8_t d[32];
        bool ok = true;
        for (int i = 0; i < 32; i++) { const long v = strtol(p, &e, 10); if (e == p) { ok = false; break; } p = e; d[i] = (uint8_t)v; }
        if (!ok) break;
        const double w = strtod(p, &e);
        if (e == p) break;
        p = e;
        T.parent.push_back((int)pid); T.desc.insert(T.desc.end(), d, d + 32); T.weight.push_back(w); T.leaf.push_back(is_leaf > 0);
    }
    if (T.parent.size() < 2) { viorb::set_error("%s holds no nodes", path); return VIORB_ERR_INVALID_ARG; }
    return VIORB_OK;
}

int read_binary(const char* path, FlatTree& T) {
    FILE* f = fopen(path, "rb");
    if (!f) { viorb::set_error("cannot open %s", path); return VIORB_ERR_INVALID_ARG; }
    uint32_t nb_nodes = 0, size_node = 0; int32_t hdr[4] = {0, 0, 0, 0};
    const bool okh = fread(&nb_nodes, 4, 1, f) == 1 && fread(&size_node, 4, 1, f) == 1 && fread(hdr, 4, 4, f) == 4;
    if (!okh || size_node < 41 || size_node > 4096 || nb_nodes < 2 || nb_nodes > (1u << 26)) {
        fclose(f); viorb::set_error("%s is not a vocabulary binary file", path); return VIORB_ERR_INVALID_ARG;
    }
    T.k = hdr[0]; T.L = hdr[1]; T.scoring = hdr[2]; T.weighting = hdr[3];
    if (T.L < 1 || T.L > 32) { fclose(f); viorb::set_error("%s: L = %d", path, T.L); return VIORB_ERR_INVALID_ARG; }
    T.parent.push_back(-1); T.desc.assign(32, 0); T.weight.push_back(0.0); T.leaf.push_back(0);
    std::vector<char> buf(size_node);
    for (uint32_t i = 1; i < nb_nodes; i++) {
        if (fread(buf.data(), size_node, 1, f) != 1) break;
        int32_t parent; float w;
        memcpy(&parent, buf.data(), 4); memcpy(&w, buf.data() + 4 + 32, 4);
        T.parent.push_back(parent);
        T.desc.insert(T.desc.end(), (const uint8_t*)buf.data() + 4, (const uint8_t*)buf.data() + 36);
        T.weight.push_back((double)w); T.leaf.push_back(buf[8 + 32] != 0);
    }
    fclose(f);
    if (T.parent.size() != nb_nodes) { viorb::set_error("%s is truncated: %zu of %u nodes", path, T.parent.size(), nb_nodes); return VIORB_ERR_INVALID_ARG; }
    return VIORB_OK;
}
} // namespace

extern "C" {

// Writers (saveToTextFile :1437-1460, saveToBinaryFile :1511-1533) from the flat arrays of viorb_vocabulary_create: node ids must be
// in file order (every parent before its children, children of a parent ascending), which is how the loaders number them.
static int flat_to_parent(int n_nodes, const int32_t* child_start, const int32_t* child_ids, std::vector<int>& parent) {
    parent.assign(n_nodes, -1);
    for (int n = 0; n < n_nodes; n++)
        for (int e = child_start[n]; e < child_start[n + 1]; e++) {
            const int c = child_ids[e];
            if (c <= 0 || c >= n_nodes || parent[c] >= 0) { viorb::set_error("child list is not a tree"); return VIORB_ERR_INVALID_ARG; }
            parent[c] = n;
        }
    for (int n = 1; n < n_nodes; n++) if (parent[n] < 0) { viorb::set_error("node %d has no parent", n); return VIORB_ERR_INVALID_ARG; }
    return VIORB_OK;
}

int viorb_vocabulary_save_text(const char* path, int n_nodes, int k, int L, const int32_t* child_start, const int32_t* child_ids, const uint8_t* desc,
                               const double* weight) {
    VIORB_REQUIRE(path && child_start && child_ids && desc && weight && n_nodes >= 2, "null argument");
    std::vector<int> parent;
    const int rc = flat_to_parent(n_nodes, child_start, child_ids, parent);
    if (rc != VIORB_OK) return rc;
    FILE* f = fopen(path, "wb");
    if (!f) { viorb::set_error("cannot write %s", path); return VIORB_ERR_INVALID_ARG; }
    fprintf(f, "%d %d  %d %d\n", k, L, 0, 0);                     // "m_k m_L  m_scoring m_weighting" (L1_NORM, TF_IDF)
    for (int i = 1; i < n_nodes; i++) {
        fprintf(f, "%d %d ", parent[i], child_start[i + 1] == child_start[i] ? 1 : 0);
        for (int j = 0; j < 32; j++) fprintf(f, "%d ", (int)desc[(size_t)32 * i + j]);
        fprintf(f, " %.17g\n", weight[i]);
    }
    fclose(f);
    return VIORB_OK;
}

int viorb_vocabulary_save_binary(const char* path, int n_nodes, int k, int L, const int32_t* child_start, const int32_t* child_ids, const uint8_t* desc,
                                 const double* weight) {
    VIORB_REQUIRE(path && child_start && child_ids && desc && weight && n_nodes >= 2, "null argument");
    std::vector<int> parent;
    const int rc = flat_to_parent(n_nodes, child_start, child_ids, parent);
    if (rc != VIORB_OK) return rc;
    FILE* f = fopen(path, "wb");
    if (!f) { viorb::set_error("cannot write %s", path); return VIORB_ERR_INVALID_ARG; }
    const uint32_t nb = (uint32_t)n_nodes, size_node = 4 + 32 + 4 + 1; const int32_t hdr[4] = {k, L, 0, 0};
    fwrite(&nb, 4, 1, f); fwrite(&size_node, 4, 1, f); fwrite(hdr, 4, 4, f);
    for (int i = 1; i < n_nodes; i++) {
        const int32_t p = parent[i]; const float w = (float)weight[i]; const uint8_t leaf = child_start[i + 1] == child_start[i];
        fwrite(&p, 4, 1, f); fwrite(desc + (size_t)32 * i, 32, 1, f); fwrite(&w, 4, 1, f); fwrite(&leaf, 1, 1, f);
    }
    fclose(f);
    return VIORB_OK;
}

} // extern "C"
