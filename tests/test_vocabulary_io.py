"""The two vocabulary file formats of the reference (TemplatedVocabulary::loadFromTextFile / loadFromBinaryFile,
Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1351-1508; the binary file is what tools/bin_vocabulary.cc writes): a synthetic tree is
written in both formats, parsed back (CPU) and — on the GPU — gives identical transform() results whichever way it was loaded."""
import struct
import numpy as np
import pytest
from viorb_amd.synth import make_vocabulary, descriptors_near_words


def _file_ordered(voc):
    """The synthetic tree as a loader would number it: children of every parent in ascending id (= file) order, float32 weights."""
    v = dict(voc)
    cs, ci = v["child_start"], v["child_ids"].copy()
    for n in range(len(cs) - 1):
        ci[cs[n]:cs[n + 1]] = np.sort(ci[cs[n]:cs[n + 1]])
    v["child_ids"] = ci
    v["weight"] = v["weight"].astype(np.float32).astype(np.float64)
    return v


@pytest.fixture(scope="module")
def voc():
    return _file_ordered(make_vocabulary(3, k=6, L=3))


def _literal_text(path, v):
    """saveToTextFile written from the reference text (:1437-1460), independently of the library's writer."""
    n = len(v["word_id"])
    parent = np.zeros(n, np.int64)
    for p in range(n):
        parent[v["child_ids"][v["child_start"][p]:v["child_start"][p + 1]]] = p
    with open(path, "w") as f:
        f.write("%d %d  %d %d\n" % (v["k"], v["L"], 0, 0))
        for i in range(1, n):
            leaf = v["child_start"][i + 1] == v["child_start"][i]
            f.write("%d %d " % (parent[i], 1 if leaf else 0) + "".join("%d " % b for b in v["desc"][i]) + " " + repr(float(v["weight"][i])) + "\n")


def _literal_binary(path, v):
    """saveToBinaryFile (:1511-1533)."""
    n = len(v["word_id"])
    parent = np.zeros(n, np.int64)
    for p in range(n):
        parent[v["child_ids"][v["child_start"][p]:v["child_start"][p + 1]]] = p
    with open(path, "wb") as f:
        f.write(struct.pack("<IIiiii", n, 4 + 32 + 4 + 1, v["k"], v["L"], 0, 0))
        for i in range(1, n):
            leaf = v["child_start"][i + 1] == v["child_start"][i]
            f.write(struct.pack("<i", int(parent[i])) + v["desc"][i].tobytes() + struct.pack("<f", float(v["weight"][i])) + struct.pack("<?", bool(leaf)))


def _same_tree(a, v):
    assert a["L"] == v["L"] and a["k"] == v["k"] and a["n_words"] == int((v["word_id"] >= 0).sum())
    for key in ("child_start", "child_ids", "word_id", "desc"):
        assert np.array_equal(a[key], v[key]), key
    assert np.array_equal(a["weight"][1:], v["weight"][1:])


def test_both_formats_parse_back_to_the_same_flat_tree(tmp_path, voc):
    from viorb_amd.frontend import read_vocabulary_file, write_vocabulary_file
    for binary in (False, True):
        lit = str(tmp_path / ("lit.bin" if binary else "lit.txt")); own = str(tmp_path / ("own.bin" if binary else "own.txt"))
        (_literal_binary if binary else _literal_text)(lit, voc)
        write_vocabulary_file(own, voc, binary)
        _same_tree(read_vocabulary_file(lit, binary), voc)
        _same_tree(read_vocabulary_file(own, binary), voc)
        if binary:
            assert open(lit, "rb").read() == open(own, "rb").read()      # the binary writer is byte-identical to the literal one


def test_text_loader_stops_at_the_last_complete_record_and_rejects_garbage(tmp_path, voc):
    import viorb_amd
    from viorb_amd.frontend import read_vocabulary_file
    p = str(tmp_path / "v.txt")
    _literal_text(p, voc)
    with open(p, "a") as f:
        f.write("\n\n")                                # the reference turns this into a node with an uninitialised descriptor; we do not
    _same_tree(read_vocabulary_file(p, False), voc)
    bad = str(tmp_path / "bad.txt")
    open(bad, "w").write("99 3 0 0\n0 1 " + "0 " * 32 + "1.0\n")
    with pytest.raises(viorb_amd.ViorbError):
        read_vocabulary_file(bad, False)
    trunc = str(tmp_path / "t.bin")
    _literal_binary(trunc, voc)
    data = open(trunc, "rb").read()
    open(trunc, "wb").write(data[:len(data) - 50])
    with pytest.raises(viorb_amd.ViorbError):
        read_vocabulary_file(trunc, True)
    with pytest.raises(viorb_amd.ViorbError):
        read_vocabulary_file(str(tmp_path / "missing.txt"), False)


@pytest.mark.gpu
def test_transform_is_identical_whichever_way_the_vocabulary_was_loaded(tmp_path):
    import viorb_amd
    v = _file_ordered(make_vocabulary(5, k=10, L=4))
    desc = descriptors_near_words(11, v, 1500)
    ref = viorb_amd.ORBVocabulary(v)
    w0, wt0, n0 = ref.transform_features(desc)
    for binary in (False, True):
        p = str(tmp_path / ("voc.bin" if binary else "voc.txt"))
        (_literal_binary if binary else _literal_text)(p, v)
        got = viorb_amd.ORBVocabulary.load(p)
        w, wt, n = got.transform_features(desc)
        assert np.array_equal(w, w0) and np.array_equal(n, n0) and np.array_equal(wt, wt0)
        ids, vals, node = got.transform(desc)
        ids0, vals0, node0 = ref.transform(desc)
        assert np.array_equal(ids, ids0) and np.array_equal(vals, vals0) and np.array_equal(node, node0)
