#!/usr/bin/env python
"""Golden vectors for spokennlp_amd/corpus.py: run the REFERENCE's converters (functions AST-extracted from
emnlp2023-topic_segmentation/src/preprocess_data.py -- the module itself imports nltk, which is not installed) on a synthetic
corpus written to a temp dir, and store inputs + expected output lines as data in tests/golden/corpus.json.
nltk's sent_tokenize is replaced by a deterministic stub on BOTH sides (the splitter is an argument of the restatement).
Usage: PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_corpus.py"""
import ast
import json
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/emnlp2023-topic_segmentation/src/preprocess_data.py"


def stub_sent_tokenize(p):
    out = [s.strip() for s in p.replace("? ", "?|").replace(". ", ".|").split("|")]
    return [s for s in out if s]


def load_reference():
    tree = ast.parse(open(SRC).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in (
        "tokenize_method", "process_wiki_section_subset", "merge_wiki_section", "process_wiki_section", "process_wiki_folder",
        "process_wiki727k", "process_wiki50", "process_wiki_elements")]
    ns = dict(os=os, json=json, tqdm=lambda x: x, sent_tokenize=stub_sent_tokenize, sec_flag="========")
    exec(compile(ast.Module(body=keep, type_ignores=[]), "<reference functions>", "exec"), ns)
    return ns


def words(rng, n):
    return " ".join(rng.choice(["alpha", "beta", "gamma", "delta", "topic", "city", "river", "cell", "is", "of", "the"]) for _ in range(n))


def main():
    rng = random.Random(7)
    ref = load_reference()
    files = {}
    # Wiki-727K style: a preamble line, section markers (one empty section), nested folders
    for mode, n in (("train", 3), ("dev", 2), ("test", 2)):
        for i in range(n):
            lines = ["preamble that is dropped\n"]
            for s in range(rng.randint(2, 4)):
                lines.append(f"========,{s + 1},Section {s}.\n")
                if rng.random() < 0.2:
                    continue
                lines += [words(rng, rng.randint(3, 8)) + " .  \n" for _ in range(rng.randint(1, 4))]
            files[f"wiki727k/{mode}/{'AA' if i % 2 else 'AB'}/{i:02d}/doc{i}"] = "".join(lines)
    # WikiSection style json
    for name in ("disease", "city"):
        for mode, n in (("train", 2), ("validation", 1), ("test", 1)):
            data = []
            for _ in range(n):
                text, annos = "", []
                for s in range(rng.randint(2, 3)):
                    paras = []
                    for _p in range(rng.randint(1, 3)):
                        paras.append(" ".join(words(rng, rng.randint(2, 5)).capitalize() + rng.choice([". ", "? ", ". "]) for _s in range(rng.randint(1, 3))).strip())
                    sec = "\n".join(paras) + "\n"
                    annos.append({"begin": len(text), "length": len(sec), "sectionLabel": f"{name}.sec{s}"})
                    text += sec
                data.append({"text": text, "annotations": annos})
            files[f"wikisection/wikisection_en_{name}_{mode}.json"] = json.dumps(data)
    # Elements
    seg, txt = [], []
    for d in (1, 2, 10):
        for p in range(rng.randint(3, 6)):
            seg.append(f"{d},{p + 1},title{rng.randint(0, 2)},x\n")
            txt.append(words(rng, rng.randint(4, 9)) + " \n")
    files["elements/wikielements.segmenttitles"] = "".join(seg)
    files["elements/wikielements.text"] = "".join(txt)

    expected = {}
    with tempfile.TemporaryDirectory() as tmp:
        for rel, content in files.items():
            path = os.path.join(tmp, "in", rel)
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                f.write(content)
        out = os.path.join(tmp, "out")
        cwd = os.getcwd()
        os.chdir(tmp)                                   # relative input paths: the "file" field stays machine independent
        try:
            os.makedirs("out/wiki727k"); ref["process_wiki727k"]("in/wiki727k", "out/wiki727k")
            os.makedirs("out/wiki50"); ref["process_wiki50"]("in/wiki727k/test", "out/wiki50")
            os.makedirs("out/wiki_section"); ref["process_wiki_section"]("in/wikisection", "out/wiki_section")
            os.makedirs("out/wiki_elements"); ref["process_wiki_elements"]("in/elements", "out/wiki_elements")
        finally:
            os.chdir(cwd)
        for root, _, names in os.walk(out):
            for n in names:
                p = os.path.join(root, n)
                expected[os.path.relpath(p, out)] = open(p).read().splitlines()
    path = os.path.join(ROOT, "tests", "golden", "corpus.json")
    with open(path, "w") as f:
        json.dump({"files": files, "expected": expected}, f, indent=0, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: len(v) for k, v in expected.items()})


if __name__ == "__main__":
    sys.dont_write_bytecode = True
    main()
