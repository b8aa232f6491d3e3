"""Special-token table of the reference (groma/constants.py:4-25)."""
IGNORE_INDEX = -100
DEFAULT_TOKENS = {
    'pad': "[PAD]", 'bos': "<s>", 'eos': "</s>", 'unk': "<unk>", 'sep': "<sep>",
    'boi': "<img>", 'eoi': "</img>", 'bor': "<roi>", 'eor': "</roi>", 'boe': "<p>", 'eoe': "</p>",
    'image': "<image>", 'region': "<region>", 'rbox': "<refer_box>", 'gbox': "<ground_box>",
    'rfeat': "<refer_feat>", 'ground': "[grounding]",
}
REGION_IDX_TOKENS = ['<r{}>'.format(i) for i in range(100)]


def derived_token_ids(base_vocab=32000):
    """Token ids the reference obtains from tokenizer.add_tokens(DEFAULT_TOKENS + REGION_IDX_TOKENS) on the
    Vicuna/LLaMA tokenizer (groma/train/train.py:90-91): <unk>,<s>,</s> already exist, the other 114 strings are
    appended in order (SURVEY.md §8 'Derived token-id table')."""
    new = [v for k, v in DEFAULT_TOKENS.items() if k not in ('bos', 'eos', 'unk')]
    ids = {s: base_vocab + i for i, s in enumerate(new)}
    for i, s in enumerate(REGION_IDX_TOKENS):
        ids[s] = base_vocab + len(new) + i
    ids.update({"<unk>": 0, "<s>": 1, "</s>": 2})
    return ids


class SyntheticTokenizer:
    """Minimal stand-in with the two members GromaModel.init_special_token_id uses (groma/model/groma.py:136-144)."""

    def __init__(self, base_vocab=32000):
        self._ids = derived_token_ids(base_vocab)
        self.pad_token_id = self._ids[DEFAULT_TOKENS['pad']]
        self.bos_token_id, self.eos_token_id = 1, 2

    def convert_tokens_to_ids(self, tokens):
        return [self._ids[t] for t in tokens]

    def __len__(self):
        return max(self._ids.values()) + 1
