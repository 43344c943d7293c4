"""Config helpers with the reference's ``aux_func`` surface (/root/reference/aux_func.py:6-103).

Same functions, argument names, printed messages and interactive behaviour: configurations
are flat dicts stored as JSON (``hp/*.json`` keeps the reference's exact key set).
"""
import json
import os
import pickle


def make_dir_path(path='./algo1/algo2', verbose=True):
    """aux_func.py:6-15 -- create every missing directory along ``path``."""
    parts = path.replace('\\', '/').split('/')
    for i in range(1, len(parts) + 1):
        p = '/'.join(parts[:i])
        if p and not os.path.exists(p):
            if verbose:
                print(' - make_dir_path: Creando:', p)
            os.mkdir(p)
    return None


def show_diff(cfg_d, old_cfg_d, i_level=0):
    """aux_func.py:18-41 -- print the keys that differ between two dicts; returns the count."""
    pad = i_level * '    '
    n_changes = 0
    for k in sorted(set(cfg_d) | set(old_cfg_d)):
        in_new, in_old = k in cfg_d, k in old_cfg_d
        if in_new and in_old:
            if cfg_d[k] == old_cfg_d[k]:
                continue
            if type(cfg_d[k]) is dict and type(old_cfg_d[k]) is dict:
                print('{} |-> {:10s}'.format(pad, k))
                n_changes += show_diff(cfg_d[k], old_cfg_d[k], i_level + 1)
            else:
                print('{} |-> {:10s}: \t  {:15s} >>> {:15s} '.format(pad, k, str(old_cfg_d[k]), str(cfg_d[k])))
                n_changes += 1
        elif in_old:
            print('{} |-> {:10s}: \t  {:15s} >>> {:15s} '.format(pad, k, str(old_cfg_d[k]), 'ERASED!!'))
            n_changes += 1
        else:
            print('{} |-> {:10s}: \t  {:15s} >>> {:15s} '.format(pad, k, 'EMPTY!!', str(cfg_d[k])))
            n_changes += 1
    return n_changes


def load_cfg_d(cfg_path_name='./ds_cfg_d.txt'):
    """aux_func.py:43-50."""
    cfg_path_name = cfg_path_name.replace('\\', '/')
    with open(cfg_path_name, 'r') as f:
        print(' Restaurando:', cfg_path_name)
        return json.loads(f.read())


def save_cfg_d(cfg_d={}, cfg_path_name='./ds_cfg_d.txt'):
    """aux_func.py:53-84 -- write JSON; if the file exists with different content, show the
    diff and ask y/n on stdin before overwriting."""
    cfg_path_name = cfg_path_name.replace('\\', '/')
    make_dir_path(os.path.split(cfg_path_name)[0])
    answer = 'y'
    if os.path.exists(cfg_path_name):
        old_cfg_d = load_cfg_d(cfg_path_name)
        cfg_d = json.loads(json.dumps(cfg_d))
        answer = 'n'
        if old_cfg_d != cfg_d:
            answer = ''
            while answer not in ('y', 'n'):
                print(' El archivo "{}" ya existe, y a cambiado:'.format(cfg_path_name))
                show_diff(cfg_d, old_cfg_d)
                print(' Desea actualizar la configuracion?? (y/n) ', end='')
                answer = input()
                if answer not in ('y', 'n'):
                    print('Respuesta erronea "{}", intente nuevamente.'.format(answer))
    if answer == 'y':
        with open(cfg_path_name, 'w') as f:
            print(' Salvando:', cfg_path_name)
            f.write(json.dumps(cfg_d))
    return None


def load_obj(file_d='./file.net', verbose=True):
    """aux_func.py:88-94 -- unpickle an object this code base wrote itself (never used on
    files shipped by third parties)."""
    with open(file_d, 'br') as f:
        n = pickle.load(f)
    if verbose:
        print(' - Objeto', type(n), os.path.basename(file_d), 'leído de disco.')
    return n


def dump_obj(n, file_d='./file.net', verbose=True):
    """aux_func.py:96-101."""
    with open(file_d, 'bw') as f:
        pickle.dump(n, f)
    if verbose:
        print(' - Objeto', type(n), os.path.basename(file_d), 'salvado en disco.')
