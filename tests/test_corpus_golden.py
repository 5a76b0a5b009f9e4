"""spokennlp_amd/corpus.py (raw corpus -> jsonl converters) against the reference's own converters
(tools/gen_golden_corpus.py ran preprocess_data.py's functions on the synthetic corpus stored in tests/golden/corpus.json)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spokennlp_amd import corpus  # noqa: E402
from tools.gen_golden_corpus import stub_sent_tokenize  # noqa: E402

G = json.load(open(os.path.join(ROOT, "tests", "golden", "corpus.json")))


@pytest.fixture()
def tree(tmp_path, monkeypatch):
    for rel, content in G["files"].items():
        p = tmp_path / "in" / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(content)
    monkeypatch.chdir(tmp_path)
    return tmp_path


def lines(path):
    return open(path).read().splitlines()


def test_wiki727k_and_wiki50(tree):
    n = corpus.process_wiki727k("in/wiki727k", "out/wiki727k")
    assert n == {"test": 2, "dev": 2, "train": 3}
    corpus.process_wiki50("in/wiki727k/test", "out/wiki50")
    for rel in ("wiki727k/train.jsonl", "wiki727k/dev.jsonl", "wiki727k/test.jsonl", "wiki50/test.jsonl"):
        # document order is os.walk order (file-system dependent, as in the reference): compare as multisets
        assert sorted(lines(os.path.join("out", rel))) == sorted(G["expected"][rel]), rel
    ex = json.loads(lines("out/wiki727k/train.jsonl")[0])
    assert ex["labels"][-1] == 1 and len(ex["labels"]) == len(ex["sentences"]) and "preamble" not in " ".join(ex["sentences"])


def test_wiki_section(tree):
    corpus.process_wiki_section("in/wikisection", "out/wiki_section", sent_tokenize=stub_sent_tokenize)
    for rel, exp in G["expected"].items():
        if rel.startswith("wiki_section"):
            assert lines(os.path.join("out", rel)) == exp, rel


def test_wiki_section_needs_a_splitter(tree):
    try:
        import nltk  # noqa: F401
        pytest.skip("nltk present")
    except ImportError:
        pass
    with pytest.raises(RuntimeError):
        corpus.process_wiki_section("in/wikisection", "out/ws2")


def test_wiki_elements(tree):
    assert corpus.process_wiki_elements("in/elements", "out/wiki_elements") == 3
    assert lines("out/wiki_elements/test.jsonl") == G["expected"]["wiki_elements/test.jsonl"]


def test_converter_output_feeds_the_loader(tree):
    from spokennlp_amd import loader
    corpus.process_wiki727k("in/wiki727k", "out/wiki727k")
    docs = list(loader.read_jsonl("out/wiki727k/train.jsonl"))
    assert len(docs) == 3 and all(len(d["sentences"]) == len(d["labels"]) for d in docs)
