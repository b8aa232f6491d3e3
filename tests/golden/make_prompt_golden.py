"""Generates tests/golden/prompt_golden.json by importing the REFERENCE's conversation templates
(/root/reference/groma/data/conversation.py, pure python) in the build container.  Run here only; the GPU box and the
test-suite read the committed JSON."""
import importlib.util, json, os
spec = importlib.util.spec_from_file_location("ref_conversation", "/root/reference/groma/data/conversation.py")
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
cases = []
dialogs = {
    "one_open": [("USER", "What is in <image>?"), ("ASSISTANT", "")],
    "two_rounds": [("USER", "Describe <region>."), ("ASSISTANT", "A dog."), ("USER", "Where?"), ("ASSISTANT", None)],
    "tuple_msg": [("USER", ("look", "img", "mode")), ("ASSISTANT", "ok"), ("USER", "and?"), ("ASSISTANT", "")],
    "closed": [("USER", "hi"), ("ASSISTANT", "hello")],
}
for name in ("default", "llava", "llama_2"):
    conv = mod.conv_templates[name]
    for dn, turns in dialogs.items():
        cases.append({"template": name, "dialog": dn, "turns": [[r, list(m) if isinstance(m, tuple) else m] for r, m in turns],
                      "prompt": conv.get_prompt([(r, m) for r, m in turns])})
cases.append({"template": "simple", "dialog": "plain", "turns": ["<image>", "a cat", "<image>", "a dog"],
              "prompt": mod.conv_templates["simple"].get_prompt(["<image>", "a cat", "<image>", "a dog"])})
# the run_groma.py preamble (groma/eval/run_groma.py:64-75) rebuilt with the reference template object
conv = mod.conv_templates["llava"]
DT = {"image": "<image>", "region": "<region>"}
instruct = "Here is an image with region crops from it. " + "Image: {}. ".format(DT["image"]) + "Regions: {}.".format(DT["region"])
msgs = [(conv.roles[0], instruct), (conv.roles[1], "Thank you for the image! How can I assist you with it?"),
        (conv.roles[0], "Describe the image in details."), (conv.roles[1], "")]
cases.append({"template": "llava", "dialog": "run_groma", "query": "Describe the image in details.", "prompt": conv.get_prompt(msgs)})
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "prompt_golden.json")
json.dump(cases, open(out, "w"), indent=1)
print(out, len(cases))
