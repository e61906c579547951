"""Query/document text -> BM25 tokens: restatement of src/processing/preprocess_bm25.py:33-52.

    lower -> delete string.punctuation -> nltk.word_tokenize -> drop NLTK-English stopwords,
    numeric tokens and tokens of length <= 1 -> optional WordNetLemmatizer().lemmatize (noun)

NLTK and its data (punkt, stopwords, wordnet) are not available offline, so the three NLTK pieces
are restated here:
  * word_tokenize: once ASCII punctuation is gone, the Treebank tokenizer's remaining effects are
    whitespace splitting, separating the Unicode quotes/dashes it knows, and its contraction splits
    ("cannot" -> "can", "not", ...);
  * the stopword list is NLTK's English list (179 entries);
  * lemmatisation: WordNet's noun morphy needs the WordNet lemma index to accept a candidate; what is
    shipped instead is a lexicon of (token -> lemma) pairs observed in the reference's own
    pre-tokenised query files (data/*_bm25_preprocessed.csv, data/test_queries_bm25.csv), with the
    identity for unseen words.  PARITY: exact on the 17.7k shipped (query -> tokens) pairs
    (tests/test_tokeniser.py); unpinned for words outside that lexicon.
The evaluation path does not tokenise at all: it feeds the shipped pre-tokenised queries
(retrieval_eval.py:28-40, :366-378).
"""
from __future__ import annotations

import gzip
import json
import os
import re
import string
from typing import Dict, List, Optional

_PUNCT_TABLE = str.maketrans("", "", string.punctuation)

STOPWORDS = frozenset("""i me my myself we our ours ourselves you you're you've you'll you'd your yours yourself
yourselves he him his himself she she's her hers herself it it's its itself they them their theirs themselves what
which who whom this that that'll these those am is are was were be been being have has had having do does did doing
a an the and but if or because as until while of at by for with about against between into through during before
after above below to from up down in out on off over under again further then once here there when where why how
all any both each few more most other some such no nor not only own same so than too very s t can will just don
don't should should've now d ll m o re ve y ain aren aren't couldn couldn't didn didn't doesn doesn't hadn hadn't
hasn hasn't haven haven't isn isn't ma mightn mightn't mustn mustn't needn needn't shan shan't shouldn shouldn't
wasn wasn't weren weren't won won't wouldn wouldn't""".split())

# Treebank tokenizer: the Unicode quotes it pads with spaces (ASCII punctuation is already deleted; en/em
# dashes are NOT split: "mother–baby" stays one token in the shipped data)
_SPLIT_CHARS = re.compile("([«»“”‘’„])")
# Treebank CONTRACTIONS2 / CONTRACTIONS3 that survive apostrophe removal
_CONTRACTIONS = [
    (re.compile(r"\b(can)(not)\b"), r"\1 \2"),
    (re.compile(r"\b(gim)(me)\b"), r"\1 \2"),
    (re.compile(r"\b(gon)(na)\b"), r"\1 \2"),
    (re.compile(r"\b(got)(ta)\b"), r"\1 \2"),
    (re.compile(r"\b(lem)(me)\b"), r"\1 \2"),
    (re.compile(r"\b(wan)(na)\b"), r"\1 \2"),
]

_LEXICON: Optional[Dict[str, str]] = None
_LEXICON_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "lemma_lexicon.json.gz")


def _lexicon() -> Dict[str, str]:
    global _LEXICON
    if _LEXICON is None:
        if os.path.exists(_LEXICON_PATH):
            with gzip.open(_LEXICON_PATH, "rt", encoding="utf-8") as f:
                _LEXICON = json.load(f)
        else:
            _LEXICON = {}
    return _LEXICON


def word_tokenize(text: str) -> List[str]:
    text = _SPLIT_CHARS.sub(r" \1 ", text)
    for rx, rep in _CONTRACTIONS:
        text = rx.sub(rep, text)
    return text.split()


def lemmatize(token: str) -> str:
    return _lexicon().get(token, token)


def preprocess_text(text: str, use_lemmatization: bool = False) -> List[str]:
    if not text:
        return []
    text = text.lower().translate(_PUNCT_TABLE)
    tokens = [t for t in word_tokenize(text) if t not in STOPWORDS and not t.isnumeric() and len(t) > 1]
    if use_lemmatization:
        tokens = [lemmatize(t) for t in tokens]
    return tokens
