#!/usr/bin/env python3
"""Generate tests/golden/irv2_table.json: the layer table of the reference's Inception-ResNet-v2.

Runs ONLY in the build container.  /root/reference/inception_resnet_v2.py parses under Python 3 (it cannot be imported:
TensorFlow is absent), so its syntax tree is walked: every slim.conv2d / max_pool2d / avg_pool2d call inside block35,
block17, block8 and inception_resnet_v2_base is recorded with its variable_scope path, output width, kernel, stride and
padding ('SAME' from the arg_scope default; the `padding` variable of the base function is 'VALID' for
align_feature_maps=False, which is how e2e_tf_s2vt.py:112-116 builds the network), plain = normalizer_fn=None (a biased
linear projection), plus the slim.repeat counts / scales and the `# H x W x C` stage comments.  The fixture is this table
(data), not the reference's source."""
import ast
import json
import os
import re

REF = "/root/reference/inception_resnet_v2.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "irv2_table.json")


def lit(node, env):
    if isinstance(node, ast.Constant):
        return node.value
    if isinstance(node, ast.List):
        return [lit(e, env) for e in node.elts]
    if isinstance(node, ast.Name):
        return env.get(node.id, f"${node.id}")
    if isinstance(node, ast.IfExp):                       # `1 if use_atrous else 2` with output_stride = 16
        return lit(node.orelse, env)
    if isinstance(node, ast.Subscript) or isinstance(node, ast.Call):
        return "$input_channels"                           # net.get_shape()[3]: the residual projections restore the block's width
    return None


def walk(body, scope, env, out):
    for st in body:
        if isinstance(st, ast.With):
            sub = scope
            for item in st.items:
                c = item.context_expr
                if isinstance(c, ast.Call) and getattr(c.func, "attr", "") == "variable_scope":
                    a = c.args
                    if len(a) == 1 and isinstance(a[0], ast.Constant):
                        sub = scope + [a[0].value]
                    elif len(a) >= 2 and isinstance(a[1], ast.Constant):
                        sub = scope + [a[1].value]        # variable_scope(scope, 'Default', ...)
            walk(st.body, sub, env, out)
            continue
        if isinstance(st, ast.If):
            walk(st.body, scope, env, out)
            continue
        for node in ast.walk(st):
            if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and isinstance(node.func.value, ast.Name)
                    and node.func.value.id == "slim"):
                continue
            kind = node.func.attr
            kw = {k.arg: k.value for k in node.keywords}
            if kind == "conv2d":
                ent = {"op": "conv", "scope": "/".join(scope + [lit(kw["scope"], env)]), "out": lit(node.args[1], env),
                       "kernel": lit(node.args[2], env), "stride": lit(kw["stride"], env) if "stride" in kw else 1,
                       "padding": lit(kw["padding"], env) if "padding" in kw else "SAME",
                       "plain": "normalizer_fn" in kw}
                out.append(ent)
            elif kind in ("max_pool2d", "avg_pool2d"):
                out.append({"op": kind, "scope": "/".join(scope + [lit(kw["scope"], env)]), "kernel": lit(node.args[1], env),
                            "stride": lit(kw["stride"], env) if "stride" in kw else 1,
                            "padding": lit(kw["padding"], env) if "padding" in kw else "SAME"})
            elif kind == "repeat":
                out.append({"op": "repeat", "count": lit(node.args[1], env), "block": node.args[2].id, "scale": lit(kw["scale"], env)})


def main():
    src = open(REF).read()
    tree = ast.parse(src)
    fns = {f.name: f for f in tree.body if isinstance(f, ast.FunctionDef)}
    table = {}
    for name in ("block35", "block17", "block8"):
        out = []
        walk(fns[name].body, [], {}, out)
        table[name] = out
    base = []
    walk(fns["inception_resnet_v2_base"].body, [], {"padding": "VALID", "use_atrous": False}, base)
    for st in ast.walk(fns["inception_resnet_v2_base"]):   # the trailing `net = block8(net, activation_fn=None)`
        if isinstance(st, ast.Call) and isinstance(st.func, ast.Name) and st.func.id == "block8":
            base.append({"op": "block8_final", "activation": None if any(k.arg == "activation_fn" for k in st.keywords) else "relu"})
    table["base"] = base
    lines = src.split("\n")
    lo, hi = fns["inception_resnet_v2_base"].lineno, fns["inception_resnet_v2_base"].end_lineno
    table["stage_shape_comments"] = [[int(x) for x in m.groups()] for l in lines[lo:hi]
                                     for m in [re.match(r"\s*# (\d+) x (\d+) x (\d+)\s*(?:if output_stride == 8,)?\s*$", l)] if m]
    table["generator"] = "tools/make_irv2_table.py (ast walk of the reference's inception_resnet_v2.py:30-259)"
    json.dump(table, open(OUT, "w"), indent=0)
    print("base ops:", len(base), "convs:", sum(e["op"] == "conv" for e in base), "; blocks:", {k: len(table[k]) for k in ("block35", "block17", "block8")})
    print(table["stage_shape_comments"])


if __name__ == "__main__":
    main()
