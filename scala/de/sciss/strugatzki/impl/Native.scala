/*
 *  Native.scala -- binding of libstrugatzki_amd.so (include/strugatzki_amd.h) through the JNI shim
 *  jni/strugatzki_amd_jni.c.  Drop into src/main/scala/de/sciss/strugatzki/impl/ of Strugatzki next to the
 *  *Impl.scala files of this directory, which replace the reference's classes of the same names.
 *
 *  Shipped as source: no JVM toolchain exists where the native library is built and tested; the tested surface
 *  is the C ABI these methods forward to.
 */
package de.sciss.strugatzki
package impl

import de.sciss.processor.Processor
import de.sciss.processor.impl.ProcessorImpl

private[strugatzki] object Native {
  System.loadLibrary("strugatzki_amd_jni") // links libstrugatzki_amd.so

  /** Receives the progress of a native run; `false` aborts it. */
  trait Progress { def apply(p: Double): Boolean }

  /** GPUs to use, e.g. `-Dstrugatzki.devices=0,1,2,3,4,5,6,7`; default: GPU 0. */
  def devices: Array[Int] =
    sys.props.get("strugatzki.devices").fold(Array(0))(_.split(',').map(_.trim.toInt))

  @native def create (device: Int): Long                                                      // sgz_create
  @native def destroy(ctx: Long): Unit                                                        // sgz_destroy
  /** One context: sgz_run_correlation_xml; several: sgz_run_correlation_xml_multi. Returns `<matches>`. */
  @native def runCorrelation    (contexts: Array[Long], configXML: String, obs: Progress): String
  @native def runSegmentation   (ctx: Long, configXML: String, obs: Progress): String       // `<breaks>`
  @native def runSelfSimilarity (ctx: Long, configXML: String, obs: Progress): Unit         // writes the PNG (gray)
  @native def runExtraction     (ctx: Long, configXML: String, obs: Progress): Unit
  @native def runCrossSimilarity(ctx: Long, configXML: String, obs: Progress): Unit
  /** Returns min0, max0, min1, max1, ... */
  @native def runFeatureStats   (ctx: Long, featureFiles: Array[String], obs: Progress): Array[Double]
  /** Rows [xBegin, xEnd) of the similarity matrix of a `<selfsimilarity>` document, [x - xBegin][y] (x = left), into
    * the direct buffer `sims` (no colours; inputs, norms and geometry are the library's affair); returns the matrix
    * extent.  `xBegin == xEnd`: the extent alone (`sims` may be null).  sgz_run_selfsimilarity_rows_xml. */
  @native def selfSimilarityRows(ctx: Long, configXML: String, xBegin: Int, xEnd: Int, sims: java.nio.FloatBuffer): Int

  /** Contexts live as long as the JVM: a context keeps the database of its last search resident (the library skips
    * reading and uploading feature files it already holds, `sgz_db_tag`), and several processors that run at the
    * same time -- each on a context set of its own -- overlap on the GPU, where a search alone leaves the chip idle
    * for about half of its duration.  `-Dstrugatzki.contextSets=n` (default 4) sets of one context per requested
    * GPU exist per device list; a processor waits for a free one. */
  private val setsPerKey: Int = sys.props.get("strugatzki.contextSets").fold(4)(_.toInt max 1)
  private val pools = new java.util.concurrent.ConcurrentHashMap[Seq[Int], java.util.concurrent.LinkedBlockingDeque[Array[Long]]]()
  private val made  = new java.util.concurrent.ConcurrentHashMap[Seq[Int], java.util.concurrent.atomic.AtomicInteger]()

  sys.addShutdownHook {
    val it = pools.values.iterator()
    while (it.hasNext) { val q = it.next(); var c = q.poll(); while (c != null) { c.foreach(destroy); c = q.poll() } }
  }

  /** Runs `fun` with one context per GPU of `devs`: a set that is free now (the one used last first: its database
    * is the likeliest to be the one asked for again), a new one while fewer than `contextSets` exist, else the next
    * to be returned. */
  def withContexts[A](devs: Array[Int])(fun: Array[Long] => A): A = {
    val key: Seq[Int] = devs.toList
    pools.putIfAbsent(key, new java.util.concurrent.LinkedBlockingDeque[Array[Long]]())
    made .putIfAbsent(key, new java.util.concurrent.atomic.AtomicInteger(0))
    val free = pools.get(key)
    var ctx  = free.pollFirst()
    if (ctx == null) {
      if (made.get(key).incrementAndGet() <= setsPerKey) ctx = devs.map(create)
      else { made.get(key).decrementAndGet(); ctx = free.takeFirst() }
    }
    try fun(ctx) finally free.addFirst(ctx)
  }

  /** Progress and abort of a `ProcessorImpl` as the callback of a native run; a run that the callback aborted
    * comes back as an `IllegalStateException("aborted")`, which is turned into `Processor.Aborted`. */
  def run[A](p: ProcessorImpl[_, _], setProgress: Double => Unit, check: () => Unit)(fun: Progress => A): A = {
    val obs = new Progress {
      def apply(frac: Double): Boolean = {
        setProgress(frac)
        try { check(); true } catch { case Processor.Aborted() => false }
      }
    }
    try fun(obs) catch {
      case e: IllegalStateException if e.getMessage == "aborted" => throw Processor.Aborted()
    }
  }
}
