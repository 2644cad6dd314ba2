/* sdict.c -- read-name <-> dense id dictionary (reference sdict.h:21-25, sdict.c:27-86).
 *
 * Contract kept from the reference: ids are dense and assigned in order of first appearance; the first
 * length seen for a name wins; sd_squeeze() drops reads flagged del, renumbers the rest in order, frees
 * the dropped names, rebuilds the index and returns a calloc'ed old->new map (-1 = dropped).
 * The index itself is our own: open addressing over (hash, id) slots with linear probing, FNV-1a hash,
 * names compared through seq[id].name -- only the resulting mapping is observable.
 */
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include "miniasm_amd.h"
#include "ma_host.h"

typedef struct {
	uint32_t n_slot, n_used; /* n_slot is a power of two; 0 = index not built (bulk-filled dictionary: built on first use) */
	uint32_t *id;            /* id+1, 0 = empty */
	uint32_t *hv;            /* cached hash */
	char *arena;             /* names of a bulk fill live in ONE block (device-side ingest): freed as a whole */
	size_t arena_len, arena_cap; /* bytes in use / bytes the block has (0: arena_len) */
} sd_index_t;

static inline int in_arena(const sd_index_t *ix, const char *p) { return ix && ix->arena && p >= ix->arena && p < ix->arena + ix->arena_len; }

static inline uint32_t sd_hash_str(const char *s)
{
	uint32_t h = 2166136261u;
	for (; *s; ++s) h = (h ^ (uint8_t)*s) * 16777619u;
	return h;
}

static sd_index_t *ix_new(uint32_t n_slot)
{
	sd_index_t *ix = (sd_index_t*)calloc(1, sizeof(sd_index_t));
	ix->n_slot = n_slot;
	ix->id = (uint32_t*)calloc(n_slot, 4);
	ix->hv = (uint32_t*)malloc((size_t)n_slot * 4);
	return ix;
}

static void ix_free(sd_index_t *ix)
{
	if (!ix) return;
	free(ix->id); free(ix->hv); free(ix->arena); free(ix);
}

static void ix_insert_raw(sd_index_t *ix, uint32_t h, uint32_t id)
{
	uint32_t m = ix->n_slot - 1, s = h & m;
	while (ix->id[s]) s = (s + 1) & m;
	ix->id[s] = id + 1; ix->hv[s] = h;
	++ix->n_used;
}

static void ix_grow(sd_index_t *ix)
{
	uint32_t i, old_n = ix->n_slot, *oid = ix->id, *ohv = ix->hv;
	ix->n_slot <<= 1; ix->n_used = 0;
	ix->id = (uint32_t*)calloc(ix->n_slot, 4);
	ix->hv = (uint32_t*)malloc((size_t)ix->n_slot * 4);
	for (i = 0; i < old_n; ++i)
		if (oid[i]) ix_insert_raw(ix, ohv[i], oid[i] - 1);
	free(oid); free(ohv);
}

sdict_t *sd_init(void)
{
	sdict_t *d = (sdict_t*)calloc(1, sizeof(sdict_t));
	d->h = ix_new(1024);
	return d;
}

void sd_destroy(sdict_t *d)
{
	uint32_t i;
	if (d == 0) return;
	for (i = 0; i < d->n_seq; ++i)
		if (!in_arena((sd_index_t*)d->h, d->seq[i].name)) free(d->seq[i].name);
	ix_free((sd_index_t*)d->h);
	free(d->seq);
	free(d);
}

static void ix_table(sd_index_t *ix, uint32_t n_slot) /* (re)allocate an empty table; 0 = no table */
{
	free(ix->id); free(ix->hv);
	ix->id = 0; ix->hv = 0; ix->n_slot = n_slot; ix->n_used = 0;
	if (n_slot) {
		ix->id = (uint32_t*)calloc(n_slot, 4);
		ix->hv = (uint32_t*)malloc((size_t)n_slot * 4);
	}
}

/* (re)build the name index from seq[] (dictionaries filled in bulk, the shallow survivor view of pipeline.c) */
void ma_sd_reindex(sdict_t *d)
{
	uint32_t i, n_slot = 1024;
	sd_index_t *ix = (sd_index_t*)d->h;
	if (ix == 0) { ix = (sd_index_t*)calloc(1, sizeof(sd_index_t)); d->h = ix; }
	while (n_slot < 2 * (uint64_t)d->n_seq + 16) n_slot <<= 1;
	ix_table(ix, n_slot);
	for (i = 0; i < d->n_seq; ++i) ix_insert_raw(ix, sd_hash_str(d->seq[i].name), i);
}

/* reference sdict.c:55-66: "make sure the index exists" (sd_squeeze calls it there; non-static, so exported here too) */
void sd_hash(sdict_t *d)
{
	const sd_index_t *ix = (const sd_index_t*)d->h;
	if (ix == 0 || ix->n_slot == 0) ma_sd_reindex(d);
}

/* forget the index (it is rebuilt on the first sd_get / sd_put); the name arena, if any, stays */
void ma_sd_drop_index(sdict_t *d)
{
	sd_index_t *ix = (sd_index_t*)d->h;
	if (ix == 0) return;
	if (ix->arena) ix_table(ix, 0);
	else { ix_free(ix); d->h = 0; }
}

/* bulk fill (device-side ingest): `arena` holds n_seq NUL-terminated names back to back and becomes the dictionary's
 * property; lens[i] = first-seen length.  Replaces whatever the dictionary held. */
void ma_sd_fill(sdict_t *d, char *arena, size_t arena_len, uint32_t n_seq, const uint32_t *lens)
{
	uint32_t i;
	char *p = arena;
	sd_index_t *ix = (sd_index_t*)d->h;
	for (i = 0; i < d->n_seq; ++i)
		if (!in_arena(ix, d->seq[i].name)) free(d->seq[i].name);
	if (ix == 0) { ix = (sd_index_t*)calloc(1, sizeof(sd_index_t)); d->h = ix; }
	free(ix->arena);
	ix->arena = arena; ix->arena_len = arena_len; ix->arena_cap = arena_len;
	ix_table(ix, 0);
	d->n_seq = d->m_seq = n_seq;
	d->seq = (sd_seq_t*)realloc(d->seq, ((size_t)n_seq + 1) * sizeof(sd_seq_t));
	for (i = 0; i < n_seq; ++i) {
		sd_seq_t *q = &d->seq[i];
		q->name = p; q->len = lens[i]; q->aux = 0; q->del = 0;
		p += strlen(p) + 1;
	}
}

/* a big block the host is about to fill once (release with free()).  MA_HOST_THP=1: 2 MiB-aligned and advised to use huge pages -- a page fault per 2 MiB instead of
 * per 4 KiB when it is first written.  Opt-in only, like the tie walk's arrays (csrc/radix.hip): where the kernel compacts memory on such a fault (defrag = madvise) the
 * first touch stalls for tens of ms at unpredictable moments (round 6: the graph-heavy pass swung between 32 and 53 ms with it).  What pays without it is that the
 * block is first touched by SEVERAL threads (unitig_gfa.c: the formatter's threads copy their pieces; pipeline.c: the survivors' view) */
void *ma_big_alloc(size_t n)
{
	static int thp = -1;
	void *p = 0;
	if (thp < 0) { const char *e = getenv("MA_HOST_THP"); thp = e && atoi(e) != 0; }
	if (thp && n >= ((size_t)4 << 20)) {
		const size_t b = (n + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
		if (posix_memalign(&p, (size_t)2 << 20, b) == 0) { (void)madvise(p, b, MADV_HUGEPAGE); return p; }
	}
	return malloc(n ? n : 1);
}

/* A dictionary that is filled again from the device (the same sdict_t for the next input) keeps its two blocks when they are big enough: nothing is freed,
 * nothing is mapped, no page is touched for the first time.  Returns 1 and the blocks to fill (then call ma_sd_adopt with exactly these); 0: allocate. */
int ma_sd_recycle(sdict_t *d, size_t arena_len, uint32_t n_seq, char **arena, sd_seq_t **seq)
{
	sd_index_t *ix = (sd_index_t*)d->h;
	uint32_t i;
	if (ix == 0 || ix->arena == 0 || d->seq == 0) return 0;
	if ((ix->arena_cap ? ix->arena_cap : ix->arena_len) < arena_len || d->m_seq < n_seq) return 0;
	for (i = 0; i < d->n_seq; ++i) /* names that came by sd_put since the last fill */
		if (!in_arena(ix, d->seq[i].name)) free(d->seq[i].name);
	d->n_seq = 0;
	*arena = ix->arena; *seq = d->seq;
	return 1;
}

/* the same with the records ready-made (csrc/paf.hip: k_dict_seqs wrote them for this arena): the dictionary takes both blocks over */
void ma_sd_adopt(sdict_t *d, char *arena, size_t arena_len, uint32_t n_seq, sd_seq_t *seq)
{
	uint32_t i;
	sd_index_t *ix = (sd_index_t*)d->h;
	const int same = ix && ix->arena == arena && d->seq == seq; /* its own blocks, filled again (ma_sd_recycle): the old records are gone already */
	if (!same)
		for (i = 0; i < d->n_seq; ++i)
			if (!in_arena(ix, d->seq[i].name)) free(d->seq[i].name);
	if (ix == 0) { ix = (sd_index_t*)calloc(1, sizeof(sd_index_t)); d->h = ix; }
	if (!same) {
		free(ix->arena);
		ix->arena = arena; ix->arena_cap = arena_len;
		free(d->seq);
		d->seq = seq; d->m_seq = n_seq;
	}
	ix->arena_len = arena_len;
	ix_table(ix, 0);
	d->n_seq = n_seq;
}

int32_t sd_get(const sdict_t *d, const char *name)
{
	const sd_index_t *ix = (const sd_index_t*)d->h;
	uint32_t h = sd_hash_str(name), m, s;
	if ((ix == 0 || ix->n_slot == 0) && d->n_seq) { ma_sd_reindex((sdict_t*)d); ix = (const sd_index_t*)d->h; } /* built on first use */
	if (ix == 0 || ix->n_slot == 0) return -1;
	m = ix->n_slot - 1;
	for (s = h & m; ix->id[s]; s = (s + 1) & m)
		if (ix->hv[s] == h && strcmp(d->seq[ix->id[s] - 1].name, name) == 0) return (int32_t)(ix->id[s] - 1);
	return -1;
}

int32_t sd_put(sdict_t *d, const char *name, uint32_t len)
{
	sd_index_t *ix = (sd_index_t*)d->h;
	uint32_t h = sd_hash_str(name), m, s;
	sd_seq_t *q;
	if (ix == 0 || ix->n_slot == 0) { ma_sd_reindex(d); ix = (sd_index_t*)d->h; }
	m = ix->n_slot - 1;
	for (s = h & m; ix->id[s]; s = (s + 1) & m)
		if (ix->hv[s] == h && strcmp(d->seq[ix->id[s] - 1].name, name) == 0) return (int32_t)(ix->id[s] - 1);
	if (d->n_seq == d->m_seq) {
		d->m_seq = d->m_seq ? d->m_seq << 1 : 16;
		d->seq = (sd_seq_t*)realloc(d->seq, (size_t)d->m_seq * sizeof(sd_seq_t));
	}
	q = &d->seq[d->n_seq];
	q->name = strdup(name); q->len = len; q->aux = 0; q->del = 0;
	ix->id[s] = d->n_seq + 1; ix->hv[s] = h;
	if (++ix->n_used > (ix->n_slot >> 1) + (ix->n_slot >> 3)) ix_grow(ix);
	return (int32_t)d->n_seq++;
}

int32_t *sd_squeeze(sdict_t *d)
{
	int32_t *map = (int32_t*)calloc(d->n_seq ? d->n_seq : 1, 4);
	uint32_t i, j;
	sd_index_t *ix = (sd_index_t*)d->h;
	for (i = j = 0; i < d->n_seq; ++i) {
		if (d->seq[i].del) { if (!in_arena(ix, d->seq[i].name)) free(d->seq[i].name); map[i] = -1; }
		else { d->seq[j] = d->seq[i]; map[i] = (int32_t)j++; }
	}
	d->n_seq = j;
	ma_sd_reindex(d);
	return map;
}
