"""Expressions of the `w_i value` entries of a dflo .prm file.

dflo hands these strings to deal.II's FunctionParser (muparser) with the variables "x,y,t" for boundary
values (src/parameters.cc:441-477) and "x,y" for the initial condition (src/parameters.cc:481-493).  This is
a small recursive-descent parser of the muparser subset the shipped examples use (and a little more): numbers,
x y z t, pi e, + - * / ^, comparisons (value 1/0), && ||, unary minus, parentheses, if(c,a,b) and the usual
one- and two-argument functions.  An expression compiles to a closure evaluated with numpy on arrays.
"""
import math
import re

import numpy as np

_TOKEN = re.compile(r"\s*(?:(\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)|([A-Za-z_][A-Za-z_0-9]*)|(<=|>=|==|!=|&&|\|\||[-+*/^()<>,?:]))")

_FUNC1 = {
    "sin": np.sin, "cos": np.cos, "tan": np.tan, "asin": np.arcsin, "acos": np.arccos, "atan": np.arctan,
    "sinh": np.sinh, "cosh": np.cosh, "tanh": np.tanh, "exp": np.exp, "log": np.log, "ln": np.log,
    "log10": np.log10, "log2": np.log2, "sqrt": np.sqrt, "abs": np.abs, "sign": np.sign, "rint": np.rint,
    "floor": np.floor, "ceil": np.ceil, "int": np.trunc, "sec": lambda a: 1.0 / np.cos(a),
    "erfc": np.vectorize(math.erfc, otypes=[float]), "erf": np.vectorize(math.erf, otypes=[float]),
}
_FUNC2 = {"min": np.minimum, "max": np.maximum, "pow": np.power, "atan2": np.arctan2, "fmod": np.fmod}
_CONST = {"pi": math.pi, "Pi": math.pi, "PI": math.pi, "_pi": math.pi, "e": math.e, "_e": math.e}


class ExpressionError(ValueError):
    pass


def _tokenize(text):
    pos, out = 0, []
    text = text.strip()
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise ExpressionError("cannot parse %r at %r" % (text, text[pos:]))
        num, name, op = m.groups()
        out.append(("num", float(num)) if num is not None else (("name", name) if name is not None else ("op", op)))
        pos = m.end()
    out.append(("end", None))
    return out


class _Parser:
    # precedence (low to high): ?:  ||  &&  == !=  < <= > >=  + -  * /  unary -  ^
    def __init__(self, text, variables):
        self.toks = _tokenize(text)
        self.i = 0
        self.variables = variables
        self.text = text

    def peek(self):
        return self.toks[self.i]

    def take(self, kind=None, val=None):
        t = self.toks[self.i]
        if (kind and t[0] != kind) or (val is not None and t[1] != val):
            raise ExpressionError("unexpected %r in %r" % (t[1], self.text))
        self.i += 1
        return t

    def is_op(self, *ops):
        t = self.peek()
        return t[0] == "op" and t[1] in ops

    def parse(self):
        f = self.ternary()
        self.take("end")
        return f

    def ternary(self):
        c = self.logic_or()
        if self.is_op("?"):
            self.take()
            a = self.ternary()
            self.take("op", ":")
            b = self.ternary()
            return lambda env: np.where(c(env) != 0, a(env), b(env))
        return c

    def _binary(self, sub, table):
        f = sub()
        while self.is_op(*table):
            op = table[self.take()[1]]
            g = sub()
            f = (lambda f, g, op: lambda env: op(f(env), g(env)))(f, g, op)
        return f

    def logic_or(self):
        return self._binary(self.logic_and, {"||": lambda a, b: ((a != 0) | (b != 0)) * 1.0})

    def logic_and(self):
        return self._binary(self.equality, {"&&": lambda a, b: ((a != 0) & (b != 0)) * 1.0})

    def equality(self):
        return self._binary(self.relational, {"==": lambda a, b: (a == b) * 1.0, "!=": lambda a, b: (a != b) * 1.0})

    def relational(self):
        return self._binary(self.additive, {"<": lambda a, b: (a < b) * 1.0, "<=": lambda a, b: (a <= b) * 1.0,
                                            ">": lambda a, b: (a > b) * 1.0, ">=": lambda a, b: (a >= b) * 1.0})

    def additive(self):
        return self._binary(self.multiplicative, {"+": lambda a, b: a + b, "-": lambda a, b: a - b})

    def multiplicative(self):
        return self._binary(self.unary, {"*": lambda a, b: a * b, "/": lambda a, b: a / b})

    def unary(self):
        if self.is_op("-"):
            self.take()
            f = self.unary()
            return lambda env: -f(env)
        if self.is_op("+"):
            self.take()
            return self.unary()
        return self.power()

    def power(self):
        base = self.atom()
        if self.is_op("^"):
            self.take()
            ex = self.unary()  # right associative, binds tighter than unary minus on its left
            return lambda env: np.power(base(env), ex(env))
        return base

    def atom(self):
        kind, val = self.peek()
        if kind == "num":
            self.take()
            return lambda env, v=val: v
        if kind == "op" and val == "(":
            self.take()
            f = self.ternary()
            self.take("op", ")")
            return f
        if kind == "name":
            self.take()
            if self.is_op("("):
                self.take()
                args = [self.ternary()]
                while self.is_op(","):
                    self.take()
                    args.append(self.ternary())
                self.take("op", ")")
                if val == "if" and len(args) == 3:
                    c, a, b = args
                    return lambda env: np.where(c(env) != 0, a(env), b(env))
                if val in _FUNC1 and len(args) == 1:
                    return lambda env, fn=_FUNC1[val], a=args[0]: fn(a(env))
                if val in _FUNC2 and len(args) == 2:
                    return lambda env, fn=_FUNC2[val], a=args[0], b=args[1]: fn(a(env), b(env))
                raise ExpressionError("unknown function %s/%d in %r" % (val, len(args), self.text))
            if val in self.variables:
                return lambda env, n=val: env[n]
            if val in _CONST:
                return lambda env, v=_CONST[val]: v
            raise ExpressionError("unknown symbol %r in %r" % (val, self.text))
        raise ExpressionError("unexpected %r in %r" % (val, self.text))


def compile_expression(text, variables=("x", "y", "t")):
    """-> f(**values) evaluating `text`; values are numpy arrays or scalars, the result broadcasts over them."""
    fn = _Parser(text, tuple(variables)).parse()

    def evaluate(**values):
        shape = np.broadcast(*[np.asarray(values[v]) for v in variables if v in values]).shape if values else ()
        r = fn({v: np.asarray(values.get(v, 0.0), dtype=np.float64) for v in variables})
        return np.broadcast_to(np.asarray(r, dtype=np.float64), shape).copy()

    return evaluate


class VectorFunction:
    """The 4-component FunctionParser of a boundary / initial-condition subsection (components w_0..w_3 =
    x-momentum, y-momentum, density, energy)."""

    def __init__(self, expressions, variables=("x", "y", "t")):
        self.expressions = list(expressions)
        self.variables = tuple(variables)
        self._fn = [compile_expression(e, variables) for e in expressions]
        self.time_dependent = "t" in variables and any(re.search(r"(?<![A-Za-z_0-9])t(?![A-Za-z_0-9(])", e) for e in expressions)

    def __call__(self, x, y, t=0.0):
        kw = {"x": x, "y": y}
        if "t" in self.variables:
            kw["t"] = t
        return [f(**kw) for f in self._fn]
