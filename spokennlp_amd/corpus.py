"""Raw corpus -> jsonl converters of the topic-segmentation path: restatement of
emnlp2023-topic_segmentation/src/preprocess_data.py:19-224 (host-side text ETL, no GPU work).

Output schema (one JSON object per line, what `datasets/*/` builders and `spokennlp_amd.loader.read_jsonl` read):
  {"sentences": [...], "labels": [...]}   label 1 = last sentence of a section, 0 = other labelled sentence, -100 = unlabelled
  (+ "file" for Wiki-727K / Wiki-50, + "section_topic_labels" / "sentence_topic_labels" for WikiSection).

WikiSection needs a sentence splitter: the reference uses `nltk.tokenize.sent_tokenize` (punkt; nltk==3.8.1 in its
requirements.txt, not vendored).  `sent_tokenize` is therefore an argument; the default imports nltk and fails loudly without it.
Pinned against the reference's own functions (AST-extracted, run on a synthetic corpus): tests/golden/corpus.json.
"""
import json
import os

SEC_FLAG = "========"        # preprocess_data.py:16


def _nltk_sent_tokenize(text):
    try:
        from nltk.tokenize import sent_tokenize
    except ImportError as e:                                    # no silent substitute: a different splitter changes the labels
        raise RuntimeError("WikiSection conversion needs nltk's punkt sentence splitter (or pass sent_tokenize=...)") from e
    return sent_tokenize(text)


# ------------------------------------------------------------------------------------------------ Wiki-727K / Wiki-50
def wiki_lines_to_example(lines, file=None):
    """preprocess_data.py:143-169: sections start at lines beginning with "========"; every other line is one sentence; the last
    sentence of every non-empty section gets label 1, the others 0.  Lines before the first marker are dropped."""
    marks = [i for i, line in enumerate(lines) if line.startswith(SEC_FLAG)]
    marks.append(len(lines))
    sentences, labels = [], []
    for a, b in zip(marks[:-1], marks[1:]):
        if a + 1 == b:
            continue
        sec = [line.strip() for line in lines[a + 1:b]]
        sentences += sec
        labels += [0] * (len(sec) - 1) + [1]
    return {"file": file, "sentences": sentences, "labels": labels}


def process_wiki_folder(folder, out_file):
    """preprocess_data.py:133-173 (os.walk order, as the reference)."""
    files = [os.path.join(root, n) for root, _, names in os.walk(folder) for n in names]
    out = []
    for path in files:
        with open(path, "r") as f:
            out.append(json.dumps(wiki_lines_to_example(f.readlines(), path)) + "\n")
    with open(out_file, "w") as f:
        f.writelines(out)
    return len(out)


def process_wiki727k(data_folder, out_folder):
    """preprocess_data.py:176-181"""
    os.makedirs(out_folder, exist_ok=True)
    return {mode: process_wiki_folder(os.path.join(data_folder, mode), os.path.join(out_folder, mode + ".jsonl"))
            for mode in ("test", "dev", "train")}


def process_wiki50(data_folder, out_folder):
    """preprocess_data.py:184-186"""
    os.makedirs(out_folder, exist_ok=True)
    return process_wiki_folder(data_folder, os.path.join(out_folder, "test.jsonl"))


# ------------------------------------------------------------------------------------------------ WikiSection
def section_sentences(sec_text, sent_tokenize=_nltk_sent_tokenize):
    """preprocess_data.py:19-31: paragraphs = non-empty lines; sentences by the splitter; the last sentence of a paragraph is
    labelled 0, the others -100, the last sentence of the section 1."""
    paragraphs = [p for p in sec_text.split("\n") if p != ""]
    sents = [sent_tokenize(p) for p in paragraphs]
    labels = [([-100] * (len(s) - 1) + [0]) if len(s) >= 1 else [] for s in sents]
    flat_s = [x for s in sents for x in s]
    flat_l = [x for l in labels for x in l]
    flat_l[-1] = 1                                               # IndexError on an empty section, as in the reference
    return flat_s, flat_l


def wikisection_example(example, sent_tokenize=_nltk_sent_tokenize):
    """preprocess_data.py:52-80: one WikiSection json entry ({"text", "annotations": [{"begin", "length", "sectionLabel"}]})"""
    text = example["text"]
    sentences, labels, sec_topics, sent_topics = [], [], [], []
    for anno in example["annotations"]:
        s, l = section_sentences(text[anno["begin"]:anno["begin"] + anno["length"]], sent_tokenize)
        sentences += s
        labels += l
        sec_topics.append(anno["sectionLabel"])
        sent_topics += [anno["sectionLabel"]] * len(s)
    return {"sentences": sentences, "labels": labels, "section_topic_labels": sec_topics, "sentence_topic_labels": sent_topics}


def process_wiki_section_subset(train_file, dev_file, test_file, out_folder, sent_tokenize=_nltk_sent_tokenize):
    """preprocess_data.py:34-101; returns mode -> list of jsonl lines"""
    os.makedirs(out_folder, exist_ok=True)
    res = {}
    for path, mode in zip((train_file, dev_file, test_file), ("train", "dev", "test")):
        with open(path, "r") as f:
            data = json.load(f)
        lines = [json.dumps(wikisection_example(ex, sent_tokenize)) + "\n" for ex in data]
        with open(os.path.join(out_folder, mode + ".jsonl"), "w") as f:
            f.writelines(lines)
        res[mode] = lines
    return res


def process_wiki_section(data_folder, out_folder, sent_tokenize=_nltk_sent_tokenize):
    """preprocess_data.py:104-130: en_disease and en_city subsets, then their concatenation"""
    subsets = {}
    for name in ("disease", "city"):
        f = [os.path.join(data_folder, f"wikisection_en_{name}_{m}.json") for m in ("train", "validation", "test")]
        subsets[name] = process_wiki_section_subset(f[0], f[1], f[2], os.path.join(out_folder, f"../wiki_section_{name}"), sent_tokenize)
    os.makedirs(out_folder, exist_ok=True)
    for mode in ("train", "dev", "test"):
        with open(os.path.join(out_folder, mode + ".jsonl"), "w") as f:
            f.writelines(subsets["disease"][mode] + subsets["city"][mode])
    return subsets


# ------------------------------------------------------------------------------------------------ Elements
def process_wiki_elements(data_folder, out_folder):
    """preprocess_data.py:189-230: wikielements.text (one paragraph per line) + wikielements.segmenttitles ("doc,para,title,...");
    a paragraph is labelled 1 when the NEXT paragraph of its document has a different title (the last paragraph always)."""
    os.makedirs(out_folder, exist_ok=True)
    with open(os.path.join(data_folder, "wikielements.segmenttitles"), "r") as f:
        seg_lines = f.readlines()
    with open(os.path.join(data_folder, "wikielements.text"), "r") as f:
        para_lines = f.readlines()
    if len(seg_lines) != len(para_lines):
        raise ValueError("wikielements.text and wikielements.segmenttitles differ in length")
    docs = {}
    for seg, para in zip(seg_lines, para_lines):
        doc_index, _, title = seg.strip().split(",")[:3]
        docs.setdefault(doc_index, []).append((title, para.strip()))
    with open(os.path.join(out_folder, "test.jsonl"), "w") as f:
        for doc_index in sorted(docs.keys()):                    # string order, as the reference
            paras = docs[doc_index]
            labels, nxt = [], ""
            for title, _ in reversed(paras):
                labels.insert(0, 1 if title != nxt else 0)
                nxt = title
            f.write(json.dumps({"sentences": [p for _, p in paras], "labels": labels}) + "\n")
    return len(docs)
