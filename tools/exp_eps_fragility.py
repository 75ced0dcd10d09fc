#!/usr/bin/env python3
"""Is a greedy policy fragile against a few random actions, or does exploration damage LEARNING?  Acts eps-greedy on fixed tables
(learning rate 0: alpha table [0], alpha_min 0) and reports how first episodes end."""
import json, sys
from collections import Counter
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
from dql_multirotor_landing_amd.config import CHECK_NAMES, Q_PAPER, training_config
from dql_multirotor_landing_amd.double_q_learning import DoubleQLearningAgent
from dql_multirotor_landing_amd.engine import Engine
tables = sys.argv[1] if len(sys.argv) > 1 else str(ROOT / "tests" / "golden" / "assets")
agent = DoubleQLearningAgent.load(Path(tables))
n = 4096
for level in (1, 4):
    for eps in (0.0, 0.01, 0.05):
        eng = Engine(training_config(level, quirks=Q_PAPER, alpha_min=0.0), n, seed=7, alpha_table=np.zeros(1))
        eng.set_tables(*agent._padded())
        first = np.full(n, -1)
        eng.train_steps(1, eps)
        for t in range(470):
            eng.train_steps(1, eps)
            d, c = eng.dones()
            new = (d != 0) & (first < 0)
            first[new] = c[new]
        qa, _, _ = eng.get_tables()
        assert np.allclose(qa.ravel(), agent._padded()[0], rtol=1e-9, atol=1e-9), "tables must not move"
        h = Counter(first.tolist())
        print(json.dumps({"level": level, "eps": eps, "outcomes": {CHECK_NAMES[k] if k >= 0 else "unfinished": v / n for k, v in sorted(h.items())}}), flush=True)
        eng.close()
