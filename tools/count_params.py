"""Parameter totals of the reference's shipped PPO and DreamerV3 models, counted
by walking the shapes their modules create (no jax needed): the size of the
flat gradient buffer that the ranks all-reduce every train step
(embodied/jax/opt.py:52-54 averages every f32 gradient leaf).

    python tools/count_params.py [--reference /root/reference]

With the reference tree present the architecture settings are READ from its
`ppo/configs.yaml` / `dreamerv3/configs.yaml` (`defaults.agent`); without it
the values recorded below (the same ones, read on 2026-09-30) are used.  What
each module creates follows, line by line:

  Linear      kernel (in, out) + bias (out)            embodied/jax/nets.py:230-251
  BlockLinear kernel (g, in/g, out/g) + bias (out)     embodied/jax/nets.py:254-281
  Conv2D      kernel (k, k, in, depth) + bias (depth)  embodied/jax/nets.py:284-323
  Norm        none: nothing; rms: scale; layer: scale + shift over the last axis
                                                       embodied/jax/nets.py:361-409
  DictEmbed   init (units) + one Linear per key (one-hot of `classes` for a
              discrete key)                            embodied/jax/nets.py:503-562
  MLP         `layers` x (Linear + Norm)               embodied/jax/nets.py:565-587
  GRU         Norm over [carry, input] + Linear to 3 * units
                                                       embodied/jax/nets.py:634-671
  MLPHead     MLP + Head (categorical: Linear to classes; mse / binary: Linear to
              1; symexp_twohot: Linear to `bins`)      embodied/jax/heads.py:16-146

PPO      ppo/agent.py:128-151 (Model), ppo/nets.py:11-80 (ImpalaEncoder)
DreamerV3 dreamerv3/agent.py:33-76 (modules of the optimizer: dyn, enc, dec, rew,
         con, pol, val -- `slowval` is a SlowModel copy, not trained),
         dreamerv3/rssm.py:16-175 (RSSM), :178-252 (Encoder), :255-377 (Decoder)
"""
import argparse
import math
import os

PPO_AGENT = {                       # ppo/configs.yaml:86-96
    'enc': {'depth': 32, 'mults': [1, 2, 2], 'outmult': 16, 'norm': 'none', 'blocks': 2},
    'recurrent': True, 'rnnact': True,
    'rnn': {'units': 1024, 'norm': 'layer'},
    'actemb': {'units': 1024},
    'policy': {'layers': 0, 'units': 1024, 'norm': 'layer'},
    'value': {'layers': 0, 'units': 1024, 'norm': 'layer', 'output': 'mse', 'bins': 255},
}
DREAMER_AGENT = {                   # dreamerv3/configs.yaml:85-102 (= size200m, :146-149)
    'rssm': {'deter': 8192, 'hidden': 1024, 'stoch': 32, 'classes': 64, 'norm': 'rms',
             'imglayers': 2, 'obslayers': 1, 'dynlayers': 1, 'absolute': False, 'blocks': 8},
    'enc': {'depth': 64, 'mults': [2, 3, 4, 4], 'layers': 3, 'units': 1024, 'norm': 'rms',
            'outer': False, 'kernel': 5, 'strided': False},
    'dec': {'depth': 64, 'mults': [2, 3, 4, 4], 'layers': 3, 'units': 1024, 'norm': 'rms',
            'outer': False, 'kernel': 5, 'bspace': 8, 'strided': False},
    'rewhead': {'layers': 1, 'units': 1024, 'norm': 'rms', 'output': 'symexp_twohot', 'bins': 255},
    'conhead': {'layers': 1, 'units': 1024, 'norm': 'rms', 'output': 'binary'},
    'policy': {'layers': 3, 'units': 1024, 'norm': 'rms'},
    'value': {'layers': 3, 'units': 1024, 'norm': 'rms', 'output': 'symexp_twohot', 'bins': 255},
}


def linear(i, o):
  return i * o + o


def block_linear(i, o, g):
  return g * (i // g) * (o // g) + o


def conv(k, i, o):
  return k * k * i * o + o


def norm(kind, n):
  return {'none': 0, 'rms': n, 'layer': 2 * n}[kind]


def mlp(i, layers, units, kind):
  total = 0
  for _ in range(layers):
    total += linear(i, units) + norm(kind, units)
    i = units
  return total, i


def head(i, cfg, outputs):
  body, i = mlp(i, cfg['layers'], cfg['units'], cfg['norm'])
  return body + linear(i, outputs)


def ppo(cfg, image, actions):
  """ppo/agent.py:128-151 on one image key of shape `image` and one discrete action key."""
  h, w, c = image
  enc = 0
  e = cfg['enc']
  for mult in e['mults']:                               # ppo/nets.py:49-62
    depth = e['depth'] * mult
    enc += conv(3, c, depth)                            # s{s}in
    h, w, c = -(-h // 2), -(-w // 2), depth             # 3x3 max pool, stride 2, 'same'
    for _ in range(e['blocks']):
      enc += 2 * (norm(e['norm'], c) + conv(3, c, c))   # n1 c1 n2 c2
  flat = h * w * c
  embed = e['outmult'] * e['depth']
  enc += norm(e['norm'], flat) + linear(flat, embed) + norm(e['norm'], embed)    # outn1 outl outn2
  parts = {'enc': enc}
  feat = embed
  if cfg['recurrent']:
    inputs = embed
    if cfg['rnnact']:
      units = cfg['actemb']['units']
      parts['actemb'] = units + linear(actions, units)  # init + Linear on the one-hot
      inputs += units
    units = cfg['rnn']['units']
    parts['rnn'] = norm(cfg['rnn']['norm'], units + inputs) + linear(units + inputs, 3 * units)
    feat = units
  parts['policy'] = head(feat, cfg['policy'], actions)
  parts['value'] = head(feat, cfg['value'], 1 if cfg['value']['output'] == 'mse' else cfg['value']['bins'])
  return parts


def dreamer(cfg, image, vector, actions):
  """dreamerv3/agent.py:33-76 on one image key (or None), `vector` proprioceptive
  inputs (0 = none) and an action of `actions` one-hot classes / dimensions."""
  r = cfg['rssm']
  deter, hidden, stoch, classes, g = r['deter'], r['hidden'], r['stoch'], r['classes'], r['blocks']
  kind = r['norm']
  # Encoder (rssm.py:178-252)
  e = cfg['enc']
  enc, tokens = 0, 0
  if vector:
    body, out = mlp(vector, e['layers'], e['units'], e['norm'])
    enc += body
    tokens += out
  if image:
    h, w, c = image
    for mult in e['mults']:
      depth = e['depth'] * mult
      enc += conv(e['kernel'], c, depth) + norm(e['norm'], depth)
      h, w, c = h // 2, w // 2, depth
    tokens += h * w * c
  # RSSM (rssm.py:16-175): _core, _observe, _prior
  dyn = 0
  for size in (deter, stoch * classes, actions):         # dynin0..2 + norms
    dyn += linear(size, hidden) + norm(kind, hidden)
  x = deter + g * 3 * hidden                             # every block sees its slice of deter + the three inputs
  for _ in range(r['dynlayers']):
    dyn += block_linear(x, deter, g) + norm(kind, deter)
    x = deter
  dyn += block_linear(x, 3 * deter, g)                   # dyngru
  x = tokens if r['absolute'] else deter + tokens
  for _ in range(r['obslayers']):
    dyn += linear(x, hidden) + norm(kind, hidden)
    x = hidden
  dyn += linear(x, stoch * classes)                      # obslogit
  x = deter
  for _ in range(r['imglayers']):
    dyn += linear(x, hidden) + norm(kind, hidden)
    x = hidden
  dyn += linear(x, stoch * classes)                      # priorlogit
  # Decoder (rssm.py:255-377)
  d = cfg['dec']
  dec = 0
  feat = deter + stoch * classes
  if vector:
    body, out = mlp(feat, d['layers'], d['units'], d['norm'])
    dec += body + linear(out, vector)                    # symlog_mse head per vector key
  if image:
    depths = [d['depth'] * m for m in d['mults']]
    factor = 2 ** (len(depths) - int(d['outer']))
    mh, mw = image[0] // factor, image[1] // factor
    shape = mh * mw * depths[-1]
    dec += block_linear(deter, shape, d['bspace'])                        # sp0
    dec += linear(stoch * classes, 2 * d['units']) + norm(d['norm'], 2 * d['units'])   # sp1 + norm
    dec += linear(2 * d['units'], shape) + norm(d['norm'], depths[-1])    # sp2, spnorm (last axis = channels)
    c = depths[-1]
    for depth in reversed(depths[:-1]):
      dec += conv(d['kernel'], c, depth) + norm(d['norm'], depth)
      c = depth
    dec += conv(d['kernel'], c, image[2])                                 # imgout
  parts = {'dyn': dyn, 'enc': enc, 'dec': dec}
  parts['rew'] = head(feat, cfg['rewhead'], cfg['rewhead']['bins'])
  parts['con'] = head(feat, cfg['conhead'], 1)
  parts['pol'] = head(feat, cfg['policy'], actions)       # categorical; bounded_normal: mean + stddev = 2 * actions
  parts['val'] = head(feat, cfg['value'], cfg['value']['bins'])
  return parts


def from_reference(root):
  """The two `defaults.agent` blocks of the reference's yaml files, reduced to the
  fields counted here; None if the tree is not there."""
  try:
    import yaml
    with open(os.path.join(root, 'ppo', 'configs.yaml')) as f:
      p = yaml.safe_load(f)['defaults']['agent']
    with open(os.path.join(root, 'dreamerv3', 'configs.yaml')) as f:
      d = yaml.safe_load(f)['defaults']['agent']
  except Exception:
    return None, None
  pick = lambda src, ref: {k: src[k] for k in ref if k in src}
  ppo_cfg = {
      'enc': {**pick(p['enc'][p['enc']['typ']], PPO_AGENT['enc']), 'blocks': 2},     # blocks: ppo/nets.py:16
      'recurrent': p['recurrent'], 'rnnact': p['rnnact'],
      'rnn': pick(p['rnn'], PPO_AGENT['rnn']), 'actemb': pick(p['actemb'], PPO_AGENT['actemb']),
      'policy': pick(p['policy'], PPO_AGENT['policy']), 'value': pick(p['value'], PPO_AGENT['value'])}
  dr_cfg = {
      'rssm': pick(d['dyn'][d['dyn']['typ']], DREAMER_AGENT['rssm']),
      'enc': pick(d['enc'][d['enc']['typ']], DREAMER_AGENT['enc']),
      'dec': pick(d['dec'][d['dec']['typ']], DREAMER_AGENT['dec']),
      'rewhead': pick(d['rewhead'], DREAMER_AGENT['rewhead']),
      'conhead': pick(d['conhead'], DREAMER_AGENT['conhead']),
      'policy': pick(d['policy'], DREAMER_AGENT['policy']),
      'value': pick(d['value'], DREAMER_AGENT['value'])}
  return ppo_cfg, dr_cfg


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--reference', default='/root/reference')
  args = ap.parse_args()
  ppo_cfg, dr_cfg = from_reference(args.reference)
  if ppo_cfg is None:
    print('# reference tree not found: the settings recorded in this file are used')
    ppo_cfg, dr_cfg = PPO_AGENT, DREAMER_AGENT
  else:
    assert ppo_cfg == PPO_AGENT, ('ppo/configs.yaml differs from the recorded settings', ppo_cfg)
    assert dr_cfg == DREAMER_AGENT, ('dreamerv3/configs.yaml differs from the recorded settings', dr_cfg)
    print(f'# architecture settings read from {args.reference} (equal to the ones recorded in this file)')

  def show(title, parts):
    total = sum(parts.values())
    print(f'{title}: {total:,} parameters = {total * 4 / 1e6:.1f} MB of f32 gradients')
    print('    ' + ', '.join(f'{k} {v:,}' for k, v in parts.items()))
    return total

  print('PPO (ppo/configs.yaml:86-96: impala encoder, GRU 1024 with action embedding, linear policy / value heads)')
  bench = show('  BASELINE configs[1] shapes: image 84x84x4 u8, 6 actions (bench.py default)', ppo(ppo_cfg, (84, 84, 4), 6))
  show('  shipped atari_pong: 96x96x1 gray (configs.yaml:29), 18 actions (actions: all)', ppo(ppo_cfg, (96, 96, 1), 18))
  print('DreamerV3 (dreamerv3/configs.yaml:85-102, the default size = size200m)')
  show('  dmc_walker_walk vision (configs[2]): image 64x64x3, 24 proprio inputs, 6 action dims (bounded_normal: 2 x 6 outputs)',
       {**dreamer(dr_cfg, (64, 64, 3), 24, 6),
        'pol': head(dr_cfg['rssm']['deter'] + dr_cfg['rssm']['stoch'] * dr_cfg['rssm']['classes'], dr_cfg['policy'], 12)})
  show('  crafter (configs[3]): image 64x64x3, 17 actions', dreamer(dr_cfg, (64, 64, 3), 0, 17))
  print(f'bench.py --grad-numel default = {bench:,} (the first PPO line)')


if __name__ == '__main__':
  main()
