"""Query/document text -> BM25 tokens: restatement of src/processing/preprocess_bm25.py:33-52.

    lower -> delete string.punctuation -> nltk.word_tokenize -> drop NLTK-English stopwords,
    numeric tokens and tokens of length <= 1 -> optional WordNetLemmatizer().lemmatize (noun)

NLTK and its data (punkt, stopwords, wordnet) are not available offline, so the three NLTK pieces
are restated here:
  * word_tokenize: once ASCII punctuation is gone, the Treebank tokenizer's remaining effects are
    whitespace splitting, separating the Unicode quotes/dashes it knows, and its contraction splits
    ("cannot" -> "can", "not", ...);
  * the stopword list is NLTK's English list (179 entries);
  * lemmatisation: `WordNetLemmatizer().lemmatize(tok)` is WordNet's noun `morphy` -- exception list, then the
    detachment rules (s, ses->s, ves->f, xes->x, zes->z, ches->ch, shes->sh, men->man, ies->y) applied
    repeatedly, a candidate accepted when it is a WordNet noun lemma, the SHORTEST accepted candidate returned
    (which is why the shipped data holds discuss -> discus, nhs -> nh, ms -> m).  The algorithm is restated in
    `NounLemmatizer`; WordNet's lemma index itself is not available offline, so it stands on what the reference's
    own pre-tokenised query files (data/*_bm25_preprocessed.csv, data/test_queries_bm25.csv) show:
      1. a word seen there is answered from the observed (token -> lemma) pair: exact on all 17.7k shipped
         (query -> tokens) pairs;
      2. an unseen word runs morphy against the dictionary of every lemma those files show (outputs of the
         lemmatiser), with the irregular pairs they show (children -> child, criteria -> criterion, ...) as the
         exception list;
      3. an unseen word with no dictionary candidate takes its most specific detachment rule anyway when it looks
         like a plain plural (longer than 3 letters, no digit or dash, not ending in ss / us / is).
    PARITY: unpinned for words outside the shipped data.  Measured by 5-fold cross-validation over WORD TYPES
    (dictionary and exceptions built without the held-out words, tests/test_tokeniser.py): 96.8 % of held-out
    tokens (97.6 % of types) get WordNet's lemma, against 77.3 % when unseen words are left unchanged.
    `lemmatizer().counts` tallies which of the three routes answered.
The evaluation path does not tokenise at all: it feeds the shipped pre-tokenised queries
(retrieval_eval.py:28-40, :366-378).
"""
from __future__ import annotations

import gzip
import json
import os
import re
import string
from collections import Counter
from typing import Dict, Iterable, List, Optional, Tuple

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

NOUN_RULES = (("s", ""), ("ses", "s"), ("ves", "f"), ("xes", "x"), ("zes", "z"), ("ches", "ch"), ("shes", "sh"),
              ("men", "man"), ("ies", "y"))  # WordNet MORPHOLOGICAL_SUBSTITUTIONS[NOUN], in its order


def _detach(forms: Iterable[str]) -> List[str]:
    return [f[: -len(old)] + new for f in forms for old, new in NOUN_RULES if f.endswith(old)]


class NounLemmatizer:
    """WordNet noun morphy (what `WordNetLemmatizer().lemmatize(word)` runs, preprocess_bm25.py:49-50) over a lemma
    dictionary built from observed (token -> lemma) pairs instead of WordNet's index (module docstring)."""

    def __init__(self, pairs: Iterable[Tuple[str, str]]):
        self.seen: Dict[str, str] = dict(pairs)
        self.lemmas = set(self.seen.values())
        # irregular forms: the observed lemma is not reachable by the detachment rules (two rounds cover the data)
        self.exceptions = {a: b for a, b in self.seen.items()
                           if a != b and b not in _detach([a]) and b not in _detach(_detach([a]))}
        self.counts: Counter = Counter()

    def morphy(self, word: str) -> str:
        """Lemma of a word NOT among the observed pairs."""
        if word in self.exceptions:
            return self.exceptions[word]
        forms = _detach([word])
        found = [f for f in [word] + forms if f in self.lemmas]
        deeper = forms
        while not found and deeper:
            deeper = _detach(deeper)
            found = [f for f in deeper if f in self.lemmas]
        if found:
            self.counts["dictionary"] += 1
            return min(found, key=len)
        if forms and len(word) > 3 and not word.endswith(("ss", "us", "is")) \
                and not any(ch.isdigit() or ch in "-–" for ch in word):
            old, new = max(((o, n) for o, n in NOUN_RULES if word.endswith(o)), key=lambda r: len(r[0]))
            self.counts["rule"] += 1
            return word[: -len(old)] + new
        self.counts["unchanged"] += 1
        return word

    def lemmatize(self, word: str) -> str:
        hit = self.seen.get(word)
        if hit is not None:
            self.counts["observed"] += 1
            return hit
        return self.morphy(word)


_LEMMATIZER: Optional[NounLemmatizer] = None
_LEXICON_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "lemma_lexicon.json.gz")


def lemmatizer() -> NounLemmatizer:
    """The product's lemmatiser: every (token -> lemma) pair of the reference's shipped query files
    (tools/make_lemma_lexicon.py wrote them to data/lemma_lexicon.json.gz)."""
    global _LEMMATIZER
    if _LEMMATIZER is None:
        with gzip.open(_LEXICON_PATH, "rt", encoding="utf-8") as f:
            data = json.load(f)
        pairs = list(data["changed"].items()) + [(w, w) for w in data["unchanged"]]
        _LEMMATIZER = NounLemmatizer(pairs)
    return _LEMMATIZER


def word_tokenize(text: str) -> List[str]:
    text = _SPLIT_CHARS.sub(r" \1 ", text)
    for rx, rep in _CONTRACTIONS:
        text = rx.sub(rep, text)
    return text.split()


def lemmatize(token: str) -> str:
    return lemmatizer().lemmatize(token)


def preprocess_text(text: str, use_lemmatization: bool = False) -> List[str]:
    if not text:
        return []
    text = text.lower().translate(_PUNCT_TABLE)
    tokens = [t for t in word_tokenize(text) if t not in STOPWORDS and not t.isnumeric() and len(t) > 1]
    if use_lemmatization:
        lem = lemmatizer()
        tokens = [lem.lemmatize(t) for t in tokens]
    return tokens
