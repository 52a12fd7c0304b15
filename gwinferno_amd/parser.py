"""The containers ``construct_hierarchical_model`` consumes (gwinferno/pipeline/parser.py:13-31): a population
model class with the names of its hyper-parameters, and a prior class with its arguments.  Attribute names are the
reference's (its function reads ``.model``, ``.params``, ``.dist``, ...).  The YAML reader that fills them in the
reference (parser.py:48-163) is configuration plumbing and not part of this package: build the two dictionaries
directly (see ``gwinferno_amd.likelihood.construct_hierarchical_model``)."""
from dataclasses import dataclass
from typing import Any, Callable, Mapping, Sequence


@dataclass
class PopModel:
    """``model(**{p: hyper_params[f"{source_param}_{p}"] for p in params})`` builds the population distribution."""

    model: Callable[..., Any]
    params: Sequence[str]


@dataclass
class PopPrior:
    """``numpyro.sample(name, dist(**params))`` draws the hyper-parameter."""

    dist: Callable[..., Any]
    params: Mapping[str, Any]


@dataclass
class PopMixtureModel(PopModel):
    """A NumPyro mixture of component distributions (analysis.py:383-389); recognised and refused by this package."""

    model: Callable[..., Any]
    mixing_dist: Callable[..., Any] = None
    mixing_params: Sequence[str] = ()
    components: Sequence[Callable[..., Any]] = ()
    component_params: Sequence[Sequence[str]] = ()
    params: Sequence[str] = ()

    def __init__(self, model, mix_dist, mix_params, components, component_params):
        self.model, self.mixing_dist, self.mixing_params = model, mix_dist, mix_params
        self.components, self.component_params, self.params = components, component_params, ()
